"""Ate pairing on BLS12-381 / BN254 in plain Python -- oracle (independent acceptance test).

Used only to check the Groth16 verification equation
``e(A,B) = e(alpha,beta) * e(sum x_i*gamma_abc_i, gamma) * e(C, delta)`` (what the un-vendored
``ark-groth16/src/verifier.rs`` checks via ``SNARK::verify``, trait at
``/root/reference/snark/src/lib.rs:59-80``).  Textbook construction: F_q12 = F_q[w]/(w^12 - a*w^6 + c)
with u = w^6 - k, affine Miller loop over the untwisted point, plain final exponentiation.  The
sign of the BLS parameter is ignored (gives e(P,Q)^-1 consistently), which does not affect
product-of-pairings equalities.
"""
from __future__ import annotations

from .fields import CurveParams


class Fq12:
    """F_q[w] / (w^12 = m6*w^6 + m0)."""

    def __init__(self, curve: CurveParams):
        self.p = curve.q
        if curve.bn_like:
            # u = w^6 - 9,  u^2 = -1  =>  w^12 = 18 w^6 - 82
            self.k, self.m6, self.m0 = 9, 18, -82
        else:
            # u = w^6 - 1           =>  w^12 = 2 w^6 - 2
            self.k, self.m6, self.m0 = 1, 2, -2
        self.one = [1] + [0] * 11
        self.zero = [0] * 12

    def add(self, a, b):
        p = self.p
        return [(x + y) % p for x, y in zip(a, b)]

    def sub(self, a, b):
        p = self.p
        return [(x - y) % p for x, y in zip(a, b)]

    def neg(self, a):
        p = self.p
        return [(-x) % p for x in a]

    def mul(self, a, b):
        p = self.p
        t = [0] * 23
        for i, x in enumerate(a):
            if x == 0:
                continue
            for j, y in enumerate(b):
                t[i + j] += x * y
        for i in range(22, 11, -1):
            c = t[i]
            if c:
                t[i - 6] += self.m6 * c
                t[i - 12] += self.m0 * c
        return [v % p for v in t[:12]]

    def sqr(self, a):
        return self.mul(a, a)

    def scalar(self, a, k):
        p = self.p
        return [(x * k) % p for x in a]

    def eq(self, a, b):
        p = self.p
        return all((x - y) % p == 0 for x, y in zip(a, b))

    def is_zero(self, a):
        return all(x % self.p == 0 for x in a)

    def inv(self, a):
        """Extended Euclid on polynomials over F_p: s with s*a == 1 mod (w^12 - m6 w^6 - m0)."""
        p = self.p

        def trim(x):
            x = [v % p for v in x]
            while len(x) > 1 and x[-1] == 0:
                x.pop()
            return x

        def divmod_(n, d):
            n = list(n)
            dl = len(d) - 1
            dinv = pow(d[-1], -1, p)
            q = [0] * max(1, len(n) - dl)
            for i in range(len(n) - 1 - dl, -1, -1):
                c = n[i + dl] * dinv % p
                q[i] = c
                if c:
                    for j, dv in enumerate(d):
                        n[i + j] = (n[i + j] - c * dv) % p
            return trim(q), trim(n[:dl] if dl else [0])

        def mulsub(s0, q, s1):      # s0 - q*s1
            out = list(s0) + [0] * max(0, len(q) + len(s1) - len(s0))
            for i, x in enumerate(q):
                if x:
                    for j, y in enumerate(s1):
                        out[i + j] = (out[i + j] - x * y) % p
            return trim(out)

        mod = [(-self.m0) % p] + [0] * 5 + [(-self.m6) % p] + [0] * 5 + [1]
        r0, r1 = mod, trim(a)
        s0, s1 = [0], [1]
        if r1 == [0]:
            raise ZeroDivisionError
        while len(r1) > 1:
            q, rem = divmod_(r0, r1)
            r0, r1 = r1, rem
            s0, s1 = s1, mulsub(s0, q, s1)
        if r1[0] == 0:
            raise ZeroDivisionError
        c = pow(r1[0], -1, p)
        out = [(x * c) % p for x in s1]
        return out + [0] * (12 - len(out))

    def pow(self, a, e):
        result = self.one
        base = a
        while e:
            if e & 1:
                result = self.mul(result, base)
            base = self.sqr(base)
            e >>= 1
        return result

    def from_fq2(self, c):
        """c0 + c1*u  ->  (c0 - k*c1) + c1*w^6."""
        p = self.p
        out = [0] * 12
        out[0] = (c[0] - self.k * c[1]) % p
        out[6] = c[1] % p
        return out

    def from_fq(self, v):
        out = [0] * 12
        out[0] = v % self.p
        return out


class Pairing:
    def __init__(self, curve: CurveParams):
        self.curve = curve
        self.F = Fq12(curve)
        F = self.F
        w = [0, 1] + [0] * 10
        self.w2 = F.mul(w, w)
        self.w3 = F.mul(self.w2, w)
        self.w2i = F.inv(self.w2)
        self.w3i = F.inv(self.w3)
        self.final_exp = (curve.q ** 12 - 1) // curve.r

    def twist(self, Q):
        """G2 affine (Fq2 coords) -> point on E(F_q12)."""
        F = self.F
        nx = F.from_fq2(Q[0])
        ny = F.from_fq2(Q[1])
        if self.curve.bn_like:      # D-type twist
            return (F.mul(nx, self.w2), F.mul(ny, self.w3))
        return (F.mul(nx, self.w2i), F.mul(ny, self.w3i))   # M-type twist

    def cast_g1(self, P):
        F = self.F
        return (F.from_fq(P[0]), F.from_fq(P[1]))

    def _line(self, P1, P2, T):
        F = self.F
        x1, y1 = P1
        x2, y2 = P2
        xt, yt = T
        if not F.eq(x1, x2):
            m = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
            return F.sub(F.mul(m, F.sub(xt, x1)), F.sub(yt, y1))
        if F.eq(y1, y2):
            m = F.mul(F.scalar(F.sqr(x1), 3), F.inv(F.scalar(y1, 2)))
            return F.sub(F.mul(m, F.sub(xt, x1)), F.sub(yt, y1))
        return F.sub(xt, x1)

    def _add(self, P1, P2):
        F = self.F
        if P1 is None:
            return P2
        if P2 is None:
            return P1
        x1, y1 = P1
        x2, y2 = P2
        if F.eq(x1, x2):
            if F.eq(y1, y2):
                m = F.mul(F.scalar(F.sqr(x1), 3), F.inv(F.scalar(y1, 2)))
            else:
                return None
        else:
            m = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        nx = F.sub(F.sub(F.sqr(m), x1), x2)
        ny = F.sub(F.mul(m, F.sub(x1, nx)), y1)
        return (nx, ny)

    def miller_loop(self, Q, P):
        """Q: G2 affine (Fq2), P: G1 affine. Returns un-exponentiated F_q12 value."""
        F = self.F
        if Q is None or P is None:
            return F.one
        Qt = self.twist(Q)
        Pt = self.cast_g1(P)
        R = Qt
        f = F.one
        cnt = self.curve.ate_loop_count
        for i in range(cnt.bit_length() - 2, -1, -1):
            f = F.mul(F.sqr(f), self._line(R, R, Pt))
            R = self._add(R, R)
            if (cnt >> i) & 1:
                f = F.mul(f, self._line(R, Qt, Pt))
                R = self._add(R, Qt)
        if self.curve.bn_like:
            q = self.curve.q
            Q1 = (F.pow(Qt[0], q), F.pow(Qt[1], q))
            nQ2 = (F.pow(Q1[0], q), F.neg(F.pow(Q1[1], q)))
            f = F.mul(f, self._line(R, Q1, Pt))
            R = self._add(R, Q1)
            f = F.mul(f, self._line(R, nQ2, Pt))
        return f

    def final_exponentiate(self, f):
        return self.F.pow(f, self.final_exp)

    def pairing(self, P, Q):
        """e(P in G1, Q in G2)."""
        return self.final_exponentiate(self.miller_loop(Q, P))

    def pairing_product_is_one(self, pairs):
        """prod e(P_i, Q_i) == 1 with a single final exponentiation."""
        F = self.F
        f = F.one
        for P, Q in pairs:
            f = F.mul(f, self.miller_loop(Q, P))
        return F.eq(self.final_exponentiate(f), F.one)
