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
