"""Radix-2 evaluation domains over Fr -- oracle (test infrastructure).

Restates the published behaviour of ark-poly ``Radix2EvaluationDomain`` (un-vendored crate
``ark-poly/src/domain/radix2/{mod,fft}.rs``): natural order in and out,
``X[k] = sum_j x[j] * omega^(j*k)``; ``ifft`` includes the ``1/N`` factor; the coset domain is
``g*H`` with ``g`` the multiplicative generator of Fr (``coset_fft(x) = fft(x[k]*g^k)``,
``coset_ifft(X) = ifft(X)[k] * g^-k``).
"""
from __future__ import annotations

from .fields import CurveParams


class Domain:
    def __init__(self, curve: CurveParams, log_n: int):
        self.curve = curve
        self.r = curve.r
        self.log_n = log_n
        self.n = 1 << log_n
        self.omega = curve.root_of_unity(log_n)
        self.omega_inv = pow(self.omega, -1, self.r)
        self.n_inv = pow(self.n, -1, self.r)
        self.g = curve.fr_generator
        self.g_inv = pow(self.g, -1, self.r)

    @staticmethod
    def for_size(curve: CurveParams, min_size: int) -> "Domain":
        log_n = max(0, (max(min_size, 1) - 1).bit_length())
        return Domain(curve, log_n)

    # iterative in-place Cooley-Tukey on a bit-reversed copy
    def _transform(self, xs, w):
        n, r = self.n, self.r
        assert len(xs) == n
        a = [0] * n
        lg = self.log_n
        for i, v in enumerate(xs):
            j = int(format(i, "0%db" % lg)[::-1], 2) if lg else 0
            a[j] = v % r
        length = 2
        while length <= n:
            wl = pow(w, n // length, r)
            half = length >> 1
            tw = [1] * half
            for k in range(1, half):
                tw[k] = tw[k - 1] * wl % r
            for s in range(0, n, length):
                for k in range(half):
                    u = a[s + k]
                    v = a[s + k + half] * tw[k] % r
                    a[s + k] = (u + v) % r
                    a[s + k + half] = (u - v) % r
            length <<= 1
        return a

    def fft(self, xs):
        return self._transform(xs, self.omega)

    def ifft(self, xs):
        a = self._transform(xs, self.omega_inv)
        return [v * self.n_inv % self.r for v in a]

    def coset_fft(self, xs):
        r, g = self.r, self.g
        out, p = [], 1
        for v in xs:
            out.append(v * p % r)
            p = p * g % r
        return self.fft(out)

    def coset_ifft(self, xs):
        a = self.ifft(xs)
        r, gi = self.r, self.g_inv
        out, p = [], 1
        for v in a:
            out.append(v * p % r)
            p = p * gi % r
        return out

    def naive_dft(self, xs):
        """O(N^2) cross-check of the index convention."""
        r, n, w = self.r, self.n, self.omega
        return [sum(xs[j] * pow(w, j * k, r) for j in range(n)) % r for k in range(n)]

    def vanishing_on_coset_inv(self):
        """(g^N - 1)^-1: Z_H is constant on the coset g*H."""
        return pow(pow(self.g, self.n, self.r) - 1, -1, self.r)

    def lagrange_at(self, tau):
        """[L_k(tau)] for the domain H (ark-poly evaluate_all_lagrange_coefficients)."""
        r, n = self.r, self.n
        tau %= r
        zt = (pow(tau, n, r) - 1) % r
        if zt == 0:
            # tau in H
            out = [0] * n
            w = 1
            for k in range(n):
                if w == tau:
                    out[k] = 1
                w = w * self.omega % r
            return out
        # L_k(tau) = Z(tau)/N * omega^k / (tau - omega^k)
        dens, w = [], 1
        for k in range(n):
            dens.append((tau - w) % r)
            w = w * self.omega % r
        # batch inverse
        pref, acc = [], 1
        for d in dens:
            acc = acc * d % r
            pref.append(acc)
        inv = pow(acc, -1, r)
        invs = [0] * n
        for k in range(n - 1, -1, -1):
            prev = pref[k - 1] if k else 1
            invs[k] = inv * prev % r
            inv = inv * dens[k] % r
        c = zt * self.n_inv % r
        out, w = [], 1
        for k in range(n):
            out.append(c * w % r * invs[k] % r)
            w = w * self.omega % r
        return out
