"""Short-Weierstrass groups y^2 = x^3 + b (a = 0) over F_q (G1) and F_q2 (G2) -- oracle.

Restates the group law that ark-ec ``short_weierstrass::{Affine, Projective}`` implements
(un-vendored crate; ``ark-ec/src/models/short_weierstrass/``).  Outputs are canonical affine
points, which are unique, so any correct group law gives identical bytes.

Affine points are ``None`` (infinity) or ``(x, y)``.  Jacobian points are ``(X, Y, Z)``,
``Z == 0`` meaning infinity.
"""
from __future__ import annotations

from .fields import FpOps, Fp2Ops, CurveParams


class Group:
    def __init__(self, F, b, gen, order, name):
        self.F = F
        self.b = b
        self.gen = gen
        self.order = order
        self.name = name

    # ---- predicates -------------------------------------------------------------------------
    def is_on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.eq(F.sqr(y), F.add(F.mul(F.sqr(x), x), self.b))

    # ---- affine ops (slow, used for reference-quality checks) ---------------------------------
    def neg(self, P):
        if P is None:
            return None
        return (P[0], self.F.neg(P[1]))

    def add(self, P, Q):
        F = self.F
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if F.eq(x1, x2):
            if F.eq(y1, y2):
                if F.is_zero(y1):
                    return None
                lam = F.mul(F.mul_small(F.sqr(x1), 3), F.inv(F.mul_small(y1, 2)))
            else:
                return None
        else:
            lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        x3 = F.sub(F.sub(F.sqr(lam), x1), x2)
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    # ---- Jacobian ops --------------------------------------------------------------------------
    def to_jac(self, P):
        F = self.F
        if P is None:
            return (F.one, F.one, F.zero)
        return (P[0], P[1], F.one)

    def to_affine(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def jdouble(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z) or F.is_zero(Y):
            return (F.one, F.one, F.zero)
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        D = F.mul_small(F.sub(F.sub(F.sqr(F.add(X, B)), A), C), 2)
        E = F.mul_small(A, 3)
        Fq = F.sqr(E)
        X3 = F.sub(Fq, F.mul_small(D, 2))
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.mul_small(C, 8))
        Z3 = F.mul_small(F.mul(Y, Z), 2)
        return (X3, Y3, Z3)

    def jadd(self, J1, J2):
        F = self.F
        X1, Y1, Z1 = J1
        X2, Y2, Z2 = J2
        if F.is_zero(Z1):
            return J2
        if F.is_zero(Z2):
            return J1
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(Y1, F.mul(Z2, Z2Z2))
        S2 = F.mul(Y2, F.mul(Z1, Z1Z1))
        if F.eq(U1, U2):
            if F.eq(S1, S2):
                return self.jdouble(J1)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, U1)
        Rr = F.sub(S2, S1)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.sqr(Rr), HHH), F.mul_small(V, 2))
        Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def jadd_affine(self, J, P):
        if P is None:
            return J
        return self.jadd(J, (P[0], P[1], self.F.one))

    def jmul(self, J, k):
        F = self.F
        k = k % self.order if k >= 0 else (-k) % self.order
        acc = (F.one, F.one, F.zero)
        for bit in bin(k)[2:] if k else "":
            acc = self.jdouble(acc)
            if bit == "1":
                acc = self.jadd(acc, J)
        return acc

    def mul(self, P, k):
        """Affine scalar multiplication k*P (k taken mod group order; negative allowed)."""
        if P is None:
            return None
        if k < 0:
            return self.mul(self.neg(P), -k)
        return self.to_affine(self.jmul(self.to_jac(P), k))

    def msm(self, bases, scalars):
        """Naive sum_i k_i * P_i -> affine. Zero scalars / infinity bases contribute nothing."""
        F = self.F
        acc = (F.one, F.one, F.zero)
        for P, k in zip(bases, scalars):
            k %= self.order
            if k == 0 or P is None:
                continue
            if k == 1:
                acc = self.jadd_affine(acc, P)
            else:
                acc = self.jadd(acc, self.jmul(self.to_jac(P), k))
        return self.to_affine(acc)

    def sum(self, pts):
        F = self.F
        acc = (F.one, F.one, F.zero)
        for P in pts:
            acc = self.jadd_affine(acc, P)
        return self.to_affine(acc)

    def batch_to_affine(self, Js):
        """Montgomery batch inversion (what ark-ec normalize_batch does)."""
        F = self.F
        prods = []
        acc = F.one
        for (_, _, Z) in Js:
            if not F.is_zero(Z):
                acc = F.mul(acc, Z)
            prods.append(acc)
        inv = F.inv(acc) if not F.is_zero(acc) else F.one
        out = [None] * len(Js)
        for i in range(len(Js) - 1, -1, -1):
            X, Y, Z = Js[i]
            if F.is_zero(Z):
                continue
            prev = prods[i - 1] if i > 0 else F.one
            zi = F.mul(inv, prev)
            inv = F.mul(inv, Z)
            zi2 = F.sqr(zi)
            out[i] = (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))
        return out

    def fixed_base_muls(self, P, scalars, window=8):
        """[k*P for k in scalars] with a windowed fixed-base table (oracle-side setup helper)."""
        F = self.F
        nbits = self.order.bit_length()
        nwin = (nbits + window - 1) // window
        # table[w][d] = d * 2^(w*window) * P  (affine)
        base = self.to_jac(P)
        jt = []
        for w in range(nwin):
            row = [(F.one, F.one, F.zero)]
            cur = (F.one, F.one, F.zero)
            for d in range(1, 1 << window):
                cur = self.jadd(cur, base)
                row.append(cur)
            jt.append(row)
            for _ in range(window):
                base = self.jdouble(base)
        flat = self.batch_to_affine([p for row in jt for p in row])
        table = [flat[w * (1 << window):(w + 1) * (1 << window)] for w in range(nwin)]
        outs = []
        mask = (1 << window) - 1
        for k in scalars:
            k %= self.order
            acc = (F.one, F.one, F.zero)
            w = 0
            while k:
                d = k & mask
                if d:
                    acc = self.jadd_affine(acc, table[w][d])
                k >>= window
                w += 1
            outs.append(acc)
        return self.batch_to_affine(outs)


_cache = {}


def g1(curve: CurveParams) -> Group:
    key = (curve.name, 1)
    if key not in _cache:
        _cache[key] = Group(FpOps(curve.q), curve.g1_b, curve.g1_gen, curve.r, curve.name + ".G1")
    return _cache[key]


def g2(curve: CurveParams) -> Group:
    key = (curve.name, 2)
    if key not in _cache:
        _cache[key] = Group(Fp2Ops(curve.q), curve.g2_b, curve.g2_gen, curve.r, curve.name + ".G2")
    return _cache[key]
