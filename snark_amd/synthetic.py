"""Synthetic R1CS instances for benchmarks, built directly in CSR.

S2 "mulchain" (primary): witnesses w_0, w_1 pseudo-random; constraint i: (w_i + w_{i+1}) * w_{i+1} = w_{i+2}
for i < n-1; last constraint w_n * 1 = x_1 (public).  ell = 2, w = n + 1, nnz(A) = 2n-1, nnz(B) = nnz(C) = n.
S1 "dummy": the reference's DummyCircuit shape (/root/reference/relations/src/sr1cs/mod.rs:276-319):
n-1 copies of a*b = c plus one empty constraint; every extra witness equals a.
Values come from a splitmix64 stream (4 LE limbs mod r), seed 0x355 by default.
"""
from __future__ import annotations

import numpy as np

from .groth16 import R1CS
from .params import Curve

_M = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M
        return z ^ (z >> 31)

    def next_fr(self, r):
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % r


def _ones_coeff(curve: Curve, nnz):
    one = np.frombuffer(curve.fr_mont(1), dtype=np.uint8)
    return np.tile(one, nnz).tobytes()


def mulchain(curve: Curve, n: int, seed: int = 0x355):
    """Returns (R1CS, z as list of canonical ints)."""
    r = curve.r
    rng = SplitMix64(seed)
    vals = [rng.next_fr(r), rng.next_fr(r)]
    for i in range(n - 1):
        vals.append((vals[i] + vals[i + 1]) * vals[i + 1] % r)
    ell = 2
    z = [1, vals[n]] + vals
    w = n + 1
    idx = np.arange(n - 1, dtype=np.uint32)
    # A: rows i<n-1: cols (ell+i, ell+i+1); last row: (ell+n)
    a_col = np.empty(2 * (n - 1) + 1, dtype=np.uint32)
    a_col[0:2 * (n - 1):2] = ell + idx
    a_col[1:2 * (n - 1):2] = ell + idx + 1
    a_col[-1] = ell + n
    a_rp = np.concatenate([np.arange(0, 2 * (n - 1) + 1, 2, dtype=np.uint64), np.array([2 * (n - 1) + 1], dtype=np.uint64)])
    b_col = np.concatenate([ell + idx + 1, np.array([0], dtype=np.uint32)]).astype(np.uint32)
    c_col = np.concatenate([ell + idx + 2, np.array([1], dtype=np.uint32)]).astype(np.uint32)
    rp1 = np.arange(0, n + 1, dtype=np.uint64)
    r1 = R1CS(curve, n, ell, w, (a_rp, rp1, rp1.copy()), (a_col, b_col, c_col),
              (_ones_coeff(curve, a_col.size), _ones_coeff(curve, n), _ones_coeff(curve, n)),
              ([1] * a_col.size, [1] * n, [1] * n))
    return r1, z


def dummy(curve: Curve, n: int, a: int = 3, b: int = 5):
    """DummyCircuit with num_variables = num_constraints = n."""
    r = curve.r
    ell = 2
    w = n - 1               # a, b, and n-3 copies of a
    z = [1, a * b % r, a % r, b % r] + [a % r] * (n - 3)
    rp = np.concatenate([np.arange(0, n, dtype=np.uint64), np.array([n - 1], dtype=np.uint64)])
    a_col = np.full(n - 1, ell + 0, dtype=np.uint32)
    b_col = np.full(n - 1, ell + 1, dtype=np.uint32)
    c_col = np.full(n - 1, 1, dtype=np.uint32)
    r1 = R1CS(curve, n, ell, w, (rp, rp.copy(), rp.copy()), (a_col, b_col, c_col),
              tuple(_ones_coeff(curve, n - 1) for _ in range(3)), ([1] * (n - 1),) * 3)
    return r1, z


def z_to_mont_bytes(curve: Curve, z):
    R = 1 << 256
    r = curve.r
    return b"".join((v * R % r).to_bytes(32, "little") for v in z)
