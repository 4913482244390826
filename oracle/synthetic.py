"""Synthetic R1CS instances of SURVEY.md section 8(d) -- oracle side (test infrastructure).

S1 "dummy"    : DummyCircuit shape (/root/reference/relations/src/sr1cs/mod.rs:276-319)
S2 "mulchain" : (w_i + w_{i+1}) * w_{i+1} = w_{i+2};  last constraint w_n * 1 = x_1
S3 "bench-LC" : LC sizes 1..=10 over the last 10 variables
                (/root/reference/relations/examples/bench.rs:13,36-56), made satisfiable by
                defining c_i as a fresh witness.
Values: splitmix64 stream -> 4 LE u64 limbs -> integer mod r.
"""
from __future__ import annotations

from . import r1cs as R

MASK64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def next_fr(self, r):
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % r


def mulchain_cs(p, n, seed=0x355):
    """S2 through the ConstraintSystem restatement (small n)."""
    cs = R.ConstraintSystem(p)
    rng = SplitMix64(seed)
    vals = [rng.next_fr(p), rng.next_fr(p)]
    for i in range(n - 1):
        vals.append((vals[i] + vals[i + 1]) * vals[i + 1] % p)
    x1 = cs.new_input_variable(lambda: vals[n])
    ws = [cs.new_witness_variable(lambda v=v: v) for v in vals]
    for i in range(n - 1):
        cs.enforce_r1cs_constraint(lambda i=i: R.LC(p) + ws[i] + ws[i + 1],
                                   lambda i=i: R.LC(p) + ws[i + 1],
                                   lambda i=i: R.LC(p) + ws[i + 2])
    cs.enforce_r1cs_constraint(lambda: R.LC(p) + ws[n], lambda: R.LC(p) + R.VAR_ONE,
                               lambda: R.LC(p) + x1)
    return cs


def mulchain_direct(p, n, seed=0x355, start=None):
    """S2 built directly (fast path for large n): returns (A, B, C, z, ell)."""
    rng = SplitMix64(seed)
    vals = [rng.next_fr(p), rng.next_fr(p)] if start is None else [start[0] % p, start[1] % p]
    for i in range(n - 1):
        vals.append((vals[i] + vals[i + 1]) * vals[i + 1] % p)
    ell = 2
    z = [1, vals[n]] + vals
    A = [[(1, ell + i), (1, ell + i + 1)] for i in range(n - 1)] + [[(1, ell + n)]]
    B = [[(1, ell + i + 1)] for i in range(n - 1)] + [[(1, 0)]]
    C = [[(1, ell + i + 2)] for i in range(n - 1)] + [[(1, 1)]]
    return A, B, C, z, ell


def dummy_cs(p, n, a=3, b=5):
    cs = R.ConstraintSystem(p)
    R.dummy_circuit(cs, a, b, n, n)
    return cs


def bench_lc_cs(p, n, seed=0x355):
    """S3: random LCs of 1..=10 terms over the 10 most recent variables."""
    cs = R.ConstraintSystem(p)
    rng = SplitMix64(seed)
    vars_ = []
    vals = {}
    for _ in range(10):
        v = rng.next_fr(p)
        var = cs.new_witness_variable(lambda v=v: v)
        vars_.append(var)
        vals[var] = v
    x = cs.new_input_variable(lambda: 7)
    vars_.append(x)
    vals[x] = 7

    def rand_lc():
        k = 1 + rng.next() % 10
        terms, acc = [], 0
        pool = vars_[-10:]
        for _ in range(k):
            c = rng.next_fr(p)
            var = pool[rng.next() % len(pool)]
            terms.append((c, var))
            acc = (acc + c * vals[var]) % p
        return terms, acc

    for _ in range(n):
        ta, va = rand_lc()
        tb, vb = rand_lc()
        cval = va * vb % p
        cvar = cs.new_witness_variable(lambda v=cval: v)
        vals[cvar] = cval
        vars_.append(cvar)
        cs.enforce_r1cs_constraint(lambda t=ta: _mk(p, t), lambda t=tb: _mk(p, t),
                                   lambda c=cvar: R.LC(p) + c)
    return cs


def _mk(p, terms):
    lc = R.LC(p)
    for c, v in terms:
        lc = lc + (c, v)
    return lc


def cs_to_instance(cs: R.ConstraintSystem):
    """(A, B, C, z, ell) from a finalized constraint system (what Groth16 consumes)."""
    cs.finalize()
    m = cs.to_matrices()[R.R1CS_PREDICATE_LABEL]
    return m[0], m[1], m[2], cs.full_assignment(), cs.num_instance_variables


# ---- large instances directly in CSR (numpy), for the O3 checks at 2^18..2^20 ---------------------------------
# Each returns (n, ell, w, mats, z) with mats = three (row_ptr u64, col u32, coeff Montgomery bytes) exactly as
# tests/helpers.csr_from_rows would produce from the row lists of the builders above (checked at small n in
# tests/test_oracle_c.py), and z the full assignment as canonical ints.
def _np():
    import numpy as np
    return np


class SplitMix64Block(SplitMix64):
    """Same stream as SplitMix64 (output k = mix(seed + k * golden)), generated in numpy blocks."""

    BLOCK = 1 << 16

    def __init__(self, seed):
        super().__init__(seed)
        self._buf, self._pos = [], 0

    def _refill(self):
        np = _np()
        with np.errstate(over="ignore"):
            k = np.arange(1, self.BLOCK + 1, dtype=np.uint64)
            z = np.uint64(self.s) + k * np.uint64(0x9E3779B97F4A7C15)
            self.s = int(z[-1])
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        self._buf, self._pos = z.tolist(), 0

    def next(self):
        if self._pos >= len(self._buf):
            self._refill()
        v = self._buf[self._pos]
        self._pos += 1
        return v

    def next_fr(self, r):
        if self._pos + 4 > len(self._buf):
            return super().next_fr(r)
        b, i = self._buf, self._pos
        self._pos = i + 4
        return (b[i] | (b[i + 1] << 64) | (b[i + 2] << 128) | (b[i + 3] << 192)) % r


def _mont_bytes(p, vals):
    R = 1 << 256
    return b"".join((v * R % p).to_bytes(32, "little") for v in vals)


def mulchain_csr(p, n, seed=0x355):
    np = _np()
    rng = SplitMix64(seed)
    vals = [rng.next_fr(p), rng.next_fr(p)]
    for i in range(n - 1):
        vals.append((vals[i] + vals[i + 1]) * vals[i + 1] % p)
    ell = 2
    z = [1, vals[n]] + vals
    idx = np.arange(n - 1, dtype=np.uint32)
    a_col = np.empty(2 * (n - 1) + 1, dtype=np.uint32)
    a_col[0:2 * (n - 1):2] = ell + idx
    a_col[1:2 * (n - 1):2] = ell + idx + 1
    a_col[-1] = ell + n
    a_rp = np.concatenate([np.arange(0, 2 * (n - 1) + 1, 2, dtype=np.uint64), np.array([2 * (n - 1) + 1], dtype=np.uint64)])
    b_col = np.concatenate([ell + idx + 1, np.array([0], dtype=np.uint32)]).astype(np.uint32)
    c_col = np.concatenate([ell + idx + 2, np.array([1], dtype=np.uint32)]).astype(np.uint32)
    rp1 = np.arange(0, n + 1, dtype=np.uint64)
    one = _mont_bytes(p, [1])
    mats = [(a_rp, a_col, one * a_col.size), (rp1, b_col, one * n), (rp1.copy(), c_col, one * n)]
    return n, ell, n + 1, mats, z


def dummy_csr(p, n, a=3, b=5):
    """DummyCircuit with num_variables = num_constraints = n (sr1cs/mod.rs:276-319)."""
    np = _np()
    ell = 2
    z = [1, a * b % p, a % p, b % p] + [a % p] * (n - 3)
    rp = np.concatenate([np.arange(0, n, dtype=np.uint64), np.array([n - 1], dtype=np.uint64)])
    one = _mont_bytes(p, [1])
    mats = [(rp, np.full(n - 1, ell + 0, dtype=np.uint32), one * (n - 1)),
            (rp.copy(), np.full(n - 1, ell + 1, dtype=np.uint32), one * (n - 1)),
            (rp.copy(), np.full(n - 1, 1, dtype=np.uint32), one * (n - 1))]
    return n, ell, n - 1, mats, z


def bench_lc_csr(p, n, seed=0x355):
    """S3 with the same random stream as bench_lc_cs; rows are the compactified LCs (sorted by column, equal
    columns merged: utils/linear_combination.rs:53-82 as applied by inline_all_lcs)."""
    np = _np()
    rng = SplitMix64Block(seed)
    ell = 2                      # One, x
    # column of the k-th allocated variable: witnesses first (10), then the input x, then one witness per constraint
    cols_pool, vals_pool = [], []
    for k in range(10):
        cols_pool.append(ell + k)
        vals_pool.append(rng.next_fr(p))
    cols_pool.append(1)
    vals_pool.append(7)
    wit_vals = vals_pool[:10]
    nw = 10

    def rand_lc():
        # terms go through the restated LinearCombination (AddAssign<(F, Variable)>, utils/linear_combination.rs:204-212,
        # including get_var_loc's linear-scan branch for short LCs, which never reports a hit: equal variables stay
        # as separate entries there) so that rows are identical to the ConstraintSystem path
        k = 1 + rng.next() % 10
        acc = 0
        lc = R.LC(p)
        pc, pv = cols_pool[-10:], vals_pool[-10:]
        for _ in range(k):
            c = rng.next_fr(p)
            j = rng.next() % 10
            col = pc[j]
            lc.add_term(c, R.instance(col) if col < ell else R.witness(col - ell))
            acc = (acc + c * pv[j]) % p
        return [(R.get_variable_index(v, ell), c) for c, v in lc.t], acc

    rows = ([], [], [])
    for _ in range(n):
        ta, va = rand_lc()
        tb, vb = rand_lc()
        cval = va * vb % p
        col = ell + nw
        nw += 1
        wit_vals.append(cval)
        cols_pool.append(col)
        vals_pool.append(cval)
        rows[0].append(ta)
        rows[1].append(tb)
        rows[2].append([(col, 1)])
    mats = []
    for M in rows:
        rp = np.zeros(n + 1, dtype=np.uint64)
        cl, cf = [], []
        for i, row in enumerate(M):
            for j, c in row:
                cl.append(j)
                cf.append(c)
            rp[i + 1] = len(cl)
        mats.append((rp, np.array(cl, dtype=np.uint32), _mont_bytes(p, cf)))
    z = [1, 7] + wit_vals
    return n, ell, nw, mats, z
