"""Parity cases shared by the CPU-emulator tier and the GPU tier: each drives the C ABI through
`snark_amd._binding.Lib` and compares with the oracle on the same seeded inputs."""
from __future__ import annotations

import random

from helpers import (fr_vec_from_mont, g1_vec_raw, g2_vec_raw, pk_load_from_oracle, r1cs_load_from_rows, z_bytes)
from oracle import groth16 as G, r1cs as R, serialize as Z, synthetic as S
from oracle.curves import g1, g2
from oracle.ntt import Domain


def ntt_case(lib, ctx, C, log_n, seed=1):
    rnd = random.Random(seed + log_n)
    n = 1 << log_n
    xs = [rnd.randrange(C.r) for _ in range(n)]
    d = Domain(C, log_n)
    data = z_bytes(C, xs)
    for inv, cos, ref in ((0, 0, d.fft), (1, 0, d.ifft), (0, 1, d.coset_fft), (1, 1, d.coset_ifft)):
        out = lib.ntt(ctx, C.curve_id, data, log_n, inv, cos)
        assert fr_vec_from_mont(C, out) == ref(xs), (C.name, log_n, inv, cos)


def ntt_roundtrip_case(lib, ctx, C, log_n, seed=2):
    """Size-independent properties: ifft(fft(x)) == x, coset likewise, and X[0] == sum x (linearity probe)."""
    rnd = random.Random(seed)
    n = 1 << log_n
    xs = [rnd.getrandbits(250) % C.r for _ in range(n)]
    data = z_bytes(C, xs)
    f = lib.ntt(ctx, C.curve_id, data, log_n, 0, 0)
    assert Z.fr_from_mont(C, f[:32]) == sum(xs) % C.r
    # X[N/2] = sum (-1)^j x_j
    half = n // 2
    assert Z.fr_from_mont(C, f[32 * half:32 * half + 32]) == (sum(xs[0::2]) - sum(xs[1::2])) % C.r
    assert lib.ntt(ctx, C.curve_id, f, log_n, 1, 0) == data
    cf = lib.ntt(ctx, C.curve_id, data, log_n, 0, 1)
    assert cf != f
    assert lib.ntt(ctx, C.curve_id, cf, log_n, 1, 1) == data


def msm_case(lib, ctx, C, group, n, seed=3, edge=True):
    rnd = random.Random(seed * 1000 + n + group)
    G_ = g1(C) if group == 1 else g2(C)
    raw = Z.g1_raw if group == 1 else Z.g2_raw
    fromraw = Z.g1_from_raw if group == 1 else Z.g2_from_raw
    sz = lib.sizes(C.curve_id)
    psz = sz["g1"] if group == 1 else sz["g2"]
    ks = [rnd.randrange(C.r) for _ in range(n)]
    pts = G_.fixed_base_muls(G_.gen, [rnd.randrange(C.r) for _ in range(n)]) if n else []
    if edge and n >= 5:
        ks[0], ks[1], ks[2] = 0, 1, C.r - 1
        ks[3] = ks[4]
    if edge and n >= 16:
        pts[7] = None                                   # base at infinity
        pts[8], ks[8] = pts[9], ks[9]                   # forces P + P in a bucket
        pts[10], ks[10] = G_.neg(pts[11]), ks[11]       # forces P + (-P)
        ks[12] = 1 << 200
        ks[13] = (1 << 255) % C.r
    out = lib.msm(ctx, C.curve_id, group, b"".join(raw(C, p) for p in pts),
                  b"".join(Z.fr_canon(C, k) for k in ks), n, psz)
    assert fromraw(C, out) == G_.msm(pts, ks), (C.name, group, n)


def resident_msm_edge_case(lib, ctx, C, group, n, to_dev, seed=11):
    """Resident bases (ark355_bases_load: per-window tables; G1 tables live in the radix-2^28 form) + msm_dev with
    the exceptional cases of the mixed addition INSIDE buckets: P + P, P + (-P), bases at infinity, runs of equal
    points, zero / one / r-1 scalars.  `to_dev(bytes) -> (ptr, keepalive)` places the scalars where the library
    expects device memory."""
    rnd = random.Random(seed * 7919 + n + group)
    G_ = g1(C) if group == 1 else g2(C)
    raw = Z.g1_raw if group == 1 else Z.g2_raw
    fromraw = Z.g1_from_raw if group == 1 else Z.g2_from_raw
    sz = lib.sizes(C.curve_id)
    psz = sz["g1"] if group == 1 else sz["g2"]
    ks = [rnd.randrange(C.r) for _ in range(n)]
    pts = G_.fixed_base_muls(G_.gen, [rnd.randrange(C.r) for _ in range(n)])
    assert n >= 24
    ks[0], ks[1], ks[2] = 0, 1, C.r - 1
    pts[3] = None                                       # base at infinity
    pts[4], ks[4] = pts[5], ks[5]                       # P + P in every window
    pts[6], ks[6] = G_.neg(pts[7]), ks[7]               # P + (-P) in every window
    for j in range(8, 13):                              # five copies: double, then keep adding the same point
        pts[j], ks[j] = pts[13], ks[13]
    pts[14], ks[14] = pts[15], (C.r - ks[15]) % C.r     # k P + (-k) P: negated digits meet the same table rows
    for j in range(16, 20):                             # small scalars: everything lands in window 0
        ks[j] = j - 15
    pts[17] = pts[16]                                   # 1*P + 2*P: different buckets, same point
    pts[19], ks[19] = pts[18], ks[18]                   # and a doubling inside the sparse window
    h = lib.bases_load(ctx, C.curve_id, group, b"".join(raw(C, p) for p in pts), n)
    try:
        ptr, keep = to_dev(b"".join(Z.fr_canon(C, k) for k in ks))
        out = lib.msm_dev(ctx, h, ptr, n, 0, psz)
        assert fromraw(C, out) == G_.msm(pts, ks), (C.name, group, n)
        # a prefix whose sum is exactly the point at infinity: P5 k + P5 k ... cancel? use the (6,7) pair alone
        ptr2, keep2 = to_dev(b"".join(Z.fr_canon(C, k) for k in [0] * 6 + ks[6:8]))
        out2 = lib.msm_dev(ctx, h, ptr2, 8, 0, psz)
        assert fromraw(C, out2) is None
    finally:
        lib.dll.ark355_bases_free(h)


def msm_known_dlog_case(lib, ctx, C, group, n, seed=4, skew=None):
    """Any size: bases s_i*G made on the device (ark355_fixed_base_mul), so MSM == (sum k_i s_i) * G."""
    rnd = random.Random(seed + n)
    G_ = g1(C) if group == 1 else g2(C)
    raw = Z.g1_raw if group == 1 else Z.g2_raw
    fromraw = Z.g1_from_raw if group == 1 else Z.g2_from_raw
    sz = lib.sizes(C.curve_id)
    psz = sz["g1"] if group == 1 else sz["g2"]
    ss = [rnd.getrandbits(64) + 1 for _ in range(n)]
    bases = lib.fixed_base_mul(ctx, C.curve_id, group, raw(C, G_.gen), b"".join(Z.fr_canon(C, s) for s in ss), n, psz)
    assert fromraw(C, bases[:psz]) == G_.mul(G_.gen, ss[0])
    if skew == "equal":
        k0 = rnd.randrange(C.r)
        ks = [k0] * n
    elif skew == "boolean":
        ks = [rnd.randrange(2) if rnd.random() < 0.9 else rnd.randrange(C.r) for _ in range(n)]
    else:
        ks = [rnd.getrandbits(255) % C.r for _ in range(n)]
    out = lib.msm(ctx, C.curve_id, group, bases, b"".join(Z.fr_canon(C, k) for k in ks), n, psz)
    expect = G_.mul(G_.gen, sum(k * s for k, s in zip(ks, ss)) % C.r)
    assert fromraw(C, out) == expect, (C.name, group, n, skew)


def r1cs_case(lib, ctx, C, A, B, Cm, z, ell):
    sz = lib.sizes(C.curve_id)
    m = len(z)
    r1 = r1cs_load_from_rows(lib, ctx, C, A, B, Cm, ell, m - ell)
    try:
        zb = z_bytes(C, z)
        az, bz, cz = lib.mat_vec(ctx, r1, zb, m, len(A), sz["fr"])
        assert fr_vec_from_mont(C, az) == R.mat_vec_mul(A, z, C.r)
        assert fr_vec_from_mont(C, bz) == R.mat_vec_mul(B, z, C.r)
        assert fr_vec_from_mont(C, cz) == R.mat_vec_mul(Cm, z, C.r)
        assert lib.is_satisfied(ctx, r1, zb, m) == R.first_unsatisfied_r1cs(A, B, Cm, z, C.r)
        for k in (ell, m - 1, m // 2):
            zbad = list(z)
            zbad[k] = (zbad[k] + 1) % C.r
            assert lib.is_satisfied(ctx, r1, z_bytes(C, zbad), m) == R.first_unsatisfied_r1cs(A, B, Cm, zbad, C.r)
        h = lib.witness_map(ctx, r1, zb, m, sz["fr"])
        assert fr_vec_from_mont(C, h) == G.witness_map(C, A, B, Cm, z, ell)
        # (an unsatisfied assignment maps to the same h as the reference's seven-transform form)
        h = lib.witness_map(ctx, r1, z_bytes(C, zbad), m, sz["fr"])
        assert fr_vec_from_mont(C, h) == G.witness_map(C, A, B, Cm, zbad, ell)
    finally:
        lib.dll.ark355_r1cs_free(r1)


def witness_map_dist_case(lib, ctx, C, A, B, Cm, z, ell, worlds):
    """The distributed witness map (every rank 1/world of each vector, three all-to-all exchanges), all ranks simulated on
    one device (ark355_witness_map_dist_sim): every coefficient equals the oracle's and the replicated map's."""
    sz = lib.sizes(C.curve_id)
    m = len(z)
    r1 = r1cs_load_from_rows(lib, ctx, C, A, B, Cm, ell, m - ell)
    try:
        zb = z_bytes(C, z)
        exp = G.witness_map(C, A, B, Cm, z, ell)
        h = lib.witness_map(ctx, r1, zb, m, sz["fr"])
        assert fr_vec_from_mont(C, h) == exp
        N = lib.dll.ark355_r1cs_domain_size(r1)
        for world in worlds:
            if 8 * world * world > N:
                with pytest_raises_code(lib, -1):
                    lib.witness_map_dist_sim(ctx, r1, zb, m, sz["fr"], world)
                continue
            hd = lib.witness_map_dist_sim(ctx, r1, zb, m, sz["fr"], world)
            assert hd == h, (C.name, N, world)
    finally:
        lib.dll.ark355_r1cs_free(r1)


class pytest_raises_code:
    """context manager: the library call fails with the given ark355 status code"""

    def __init__(self, lib, code):
        self.code = code

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        assert ev is not None and getattr(ev, "code", None) == self.code, (et, ev)
        return True


def prove_case(lib, ctx, C, A, B, Cm, z, ell, td=None, rs=((0x1234567890abcdef, 0xfedcba0987654321aabbccdd),),
               verify=False):
    sz = lib.sizes(C.curve_id)
    m = len(z)
    w = m - ell
    td = td or G.Trapdoor(tau=987654321, alpha=5, beta=7, gamma=11, delta=13)
    pk = G.setup(C, A, B, Cm, ell, m, td)
    N = 1 << pk.domain_log
    r1 = r1cs_load_from_rows(lib, ctx, C, A, B, Cm, ell, w)
    pkh = pk_load_from_oracle(lib, ctx, C, pk, ell, w, N)
    try:
        assert lib.dll.ark355_r1cs_domain_size(r1) == N
        zb = z_bytes(C, z)
        for r_, s_ in rs:
            r_, s_ = r_ % C.r, s_ % C.r
            a, b, c = lib.prove(ctx, pkh, r1, zb, m, Z.fr_canon(C, r_), Z.fr_canon(C, s_), sz)
            got = G.Proof(Z.g1_from_raw(C, a), Z.g2_from_raw(C, b), Z.g1_from_raw(C, c))
            exp = G.prove_closed_form(C, pk, z, ell, r_, s_)
            assert got == exp, (C.name, len(A), r_, s_)
            assert Z.proof_bytes(C, got) == Z.proof_bytes(C, exp)
            if verify:
                assert G.verify(C, pk.vk, z[1:ell], got)
        # SynthesisError::AssignmentMissing for a short assignment
        try:
            lib.prove(ctx, pkh, r1, zb[:-32], m - 1, Z.fr_canon(C, 1), Z.fr_canon(C, 2), sz)
            assert False, "short assignment must fail"
        except Exception as e:
            assert getattr(e, "code", None) == -16
    finally:
        lib.dll.ark355_pk_free(pkh)
        lib.dll.ark355_r1cs_free(r1)
    return pk


def check_satisfied_case(lib, ctx, C, policy, n=13):
    """Policy CHECK_SATISFIED.  Off (default): an assignment that does not satisfy the constraints is proven like any other, and
    the proof is the one the reference's algorithm yields for it (oracle O1: seven transforms, five MSMs) -- the six-transform
    witness map of the library agrees with it for ANY assignment.  On: ARK355_E_UNSATISFIABLE (SynthesisError::Unsatisfiable)
    with the index of the first unsatisfied constraint; satisfied assignments prove as before."""
    sz = lib.sizes(C.curve_id)
    A, B, Cm, z, ell = S.mulchain_direct(C.r, n)
    m = len(z)
    td = G.Trapdoor(tau=987654321, alpha=5, beta=7, gamma=11, delta=13)
    pk = G.setup(C, A, B, Cm, ell, m, td)
    r1 = r1cs_load_from_rows(lib, ctx, C, A, B, Cm, ell, m - ell)
    pkh = pk_load_from_oracle(lib, ctx, C, pk, ell, m - ell, 1 << pk.domain_log)
    try:
        zbad = list(z)
        zbad[ell + n // 2] = (zbad[ell + n // 2] + 1) % C.r
        bad = R.first_unsatisfied_r1cs(A, B, Cm, zbad, C.r)
        assert bad is not None and bad >= 0
        r_, s_ = 77, 99
        a, b, c = lib.prove(ctx, pkh, r1, z_bytes(C, zbad), m, Z.fr_canon(C, r_), Z.fr_canon(C, s_), sz)
        got = G.Proof(Z.g1_from_raw(C, a), Z.g2_from_raw(C, b), Z.g1_from_raw(C, c))
        assert got == G.prove(C, pk, A, B, Cm, zbad, ell, r_, s_), "unsatisfied assignment, policy off: the reference's (unverifiable) proof"
        assert not G.verify(C, pk.vk, zbad[1:ell], got)
        policy.setenv("ARK355_CHECK_SATISFIED", "1")
        a, b, c = lib.prove(ctx, pkh, r1, z_bytes(C, z), m, Z.fr_canon(C, r_), Z.fr_canon(C, s_), sz)
        assert G.Proof(Z.g1_from_raw(C, a), Z.g2_from_raw(C, b), Z.g1_from_raw(C, c)) == G.prove_closed_form(C, pk, z, ell, r_, s_)
        try:
            lib.prove(ctx, pkh, r1, z_bytes(C, zbad), m, Z.fr_canon(C, r_), Z.fr_canon(C, s_), sz)
            assert False, "unsatisfied assignment must be refused under CHECK_SATISFIED"
        except Exception as e:
            assert getattr(e, "code", None) == -17, e
            assert ("constraint %d " % bad) in str(e), (str(e), bad)
        # the batch entry reports the same error
        try:
            lib.prove_batch(ctx, pkh, r1, [z_bytes(C, z), z_bytes(C, zbad)], m, [Z.fr_canon(C, 1)] * 2, [Z.fr_canon(C, 2)] * 2, sz, inflight=2)
            assert False, "batch with an unsatisfied assignment must fail under CHECK_SATISFIED"
        except Exception as e:
            assert getattr(e, "code", None) == -17, e
    finally:
        lib.dll.ark355_pk_free(pkh)
        lib.dll.ark355_r1cs_free(r1)


def prove_batch_case(lib, ctx, C, count=5, n=9, inflight=3):
    """ark355_prove_batch: `count` proofs of one circuit, different witnesses and randomisers, a few in flight;
    every proof must equal the oracle's closed-form proof for the same (z, r, s)."""
    sz = lib.sizes(C.curve_id)
    td = G.Trapdoor(tau=0xabcdef0123, alpha=0x1111, beta=0x2222, gamma=0x3333, delta=0x4444)
    insts = [S.mulchain_direct(C.r, n, seed=0x355 + k) for k in range(count)]
    A, B, Cm, z0, ell = insts[0]
    m = len(z0)
    pko = G.setup(C, A, B, Cm, ell, m, td)
    N = 1 << pko.domain_log
    r1 = r1cs_load_from_rows(lib, ctx, C, A, B, Cm, ell, m - ell)
    pk = pk_load_from_oracle(lib, ctx, C, pko, ell, m - ell, N)
    try:
        rs = [(0x1234 + 7 * k, 0x9999 + 13 * k) for k in range(count)]
        zs = [z_bytes(C, inst[3]) for inst in insts]
        out = lib.prove_batch(ctx, pk, r1, zs, m, [Z.fr_canon(C, r) for r, _ in rs],
                              [Z.fr_canon(C, s) for _, s in rs], sz, inflight=inflight)
        assert len(out) == count
        for k in range(count):
            got = G.Proof(Z.g1_from_raw(C, out[k][0]), Z.g2_from_raw(C, out[k][1]), Z.g1_from_raw(C, out[k][2]))
            assert got == G.prove_closed_form(C, pko, insts[k][3], ell, rs[k][0], rs[k][1]), (C.name, k)
        # error path: short assignments -> SynthesisError::AssignmentMissing for the batch
        try:
            lib.prove_batch(ctx, pk, r1, zs, m - 1, [Z.fr_canon(C, 1)] * count, [Z.fr_canon(C, 2)] * count, sz)
            assert False, "short assignment must fail"
        except Exception as e:
            assert getattr(e, "code", None) == -16
        assert lib.prove_batch(ctx, pk, r1, [], m, [], [], sz) == []
    finally:
        lib.dll.ark355_pk_free(pk)
        lib.dll.ark355_r1cs_free(r1)


def fixed_base_case(lib, ctx, C, group, n=70, seed=21):
    """ark355_fixed_base_mul (windowed table, batched normalisation) against the oracle's C fixed-base routine: random
    scalars, 0 (infinity), 1, r - 1, single-window values, a 256-bit value beyond r, and a base at infinity."""
    from oracle.c import cbase
    rnd = random.Random(seed + group)
    raw = Z.g1_raw if group == 1 else Z.g2_raw
    G_ = g1(C) if group == 1 else g2(C)
    sz = lib.sizes(C.curve_id)
    psz = sz["g1"] if group == 1 else sz["g2"]
    ks = [rnd.randrange(C.r) for _ in range(n)]
    ks[:8] = [0, 1, C.r - 1, 255, 256, 1 << 248, (1 << 256) - 1, 0xFF00FF]
    base = raw(C, G_.mul(G_.gen, 12345))
    sb = b"".join(k.to_bytes(32, "little") for k in ks)
    assert lib.fixed_base_mul(ctx, C.curve_id, group, base, sb, len(ks), psz) == cbase.fixed_base(C, group, base, sb, len(ks))
    assert lib.fixed_base_mul(ctx, C.curve_id, group, bytes(psz), sb, 9, psz) == bytes(9 * psz)
    assert lib.fixed_base_mul(ctx, C.curve_id, group, base, b"", 0, psz) == b""


def verify_batch_case(lib, ctx, C, count=2, n=9, seed=31, oracle_pairing=True, light=False):
    """ark355_verify_batch (random linear combination: count + 3 Miller loops, one final exponentiation; multi-scalar
    sums on the device) against the oracle's pairing check: valid batches accept, a tampered proof, a wrong public input
    or a proof bound to another statement make the batch fail; single-proof verification (rho = NULL) agrees with
    oracle.groth16.verify."""
    rnd = random.Random(seed)
    td = G.Trapdoor(tau=rnd.randrange(2, C.r), alpha=3, beta=5, gamma=7, delta=11)
    A, B, Cm, z0, ell = S.mulchain_direct(C.r, n, seed=seed)
    pk = G.setup(C, A, B, Cm, ell, len(z0), td)
    vk = (Z.g1_raw(C, pk.vk.alpha_g1), Z.g2_raw(C, pk.vk.beta_g2), Z.g2_raw(C, pk.vk.gamma_g2), Z.g2_raw(C, pk.vk.delta_g2),
          g1_vec_raw(C, pk.vk.gamma_abc_g1))
    proofs, inputs, oproofs, zs = [], [], [], []
    for j in range(count):
        _, _, _, z, _ = S.mulchain_direct(C.r, n, seed=seed + 1 + j)        # same circuit, different statement
        p = G.prove_closed_form(C, pk, z, ell, rnd.randrange(C.r), rnd.randrange(C.r))
        if j == 0 and oracle_pairing:
            assert G.verify(C, pk.vk, z[1:ell], p)          # the oracle's own pairing check (slow: once)
        oproofs.append(p)
        zs.append(z)
        proofs.append((Z.g1_raw(C, p.a), Z.g2_raw(C, p.b), Z.g1_raw(C, p.c)))
        inputs.append(z_bytes(C, z[1:ell]))
    rho = [Z.fr_canon(C, rnd.randrange(1, 1 << 128)) for _ in range(count)]
    assert lib.verify_batch(ctx, C.curve_id, vk, proofs, b"".join(inputs), rho)
    for j in range(1 if light else count):
        assert lib.verify_batch(ctx, C.curve_id, vk, [proofs[j]], inputs[j])          # plain verification
    # a proof bound to another statement
    assert not lib.verify_batch(ctx, C.curve_id, vk, [proofs[0]], inputs[1])
    if oracle_pairing and not light:
        assert not G.verify(C, pk.vk, zs[1][1:ell], oproofs[0])
    # tampered C in the middle of a batch
    bad = list(proofs)
    bad[1] = (proofs[1][0], proofs[1][1], proofs[0][2])
    assert not lib.verify_batch(ctx, C.curve_id, vk, bad, b"".join(inputs), rho)
    if light:            # the emulator tier stops here (every call runs two emulated device MSMs)
        return
    # wrong public input
    wrong = list(inputs)
    wrong[-1] = z_bytes(C, [(zs[-1][1] + 1) % C.r])
    assert not lib.verify_batch(ctx, C.curve_id, vk, proofs, b"".join(wrong), rho)
    # A at infinity is not a valid proof
    inf = [(bytes(len(proofs[0][0])), proofs[0][1], proofs[0][2])]
    assert not lib.verify_batch(ctx, C.curve_id, vk, inf, inputs[0])


def resident_known_dlog_case(lib, ctx, C, group, n, to_dev, seed=21):
    """Resident bases s_i * G (made by ark355_fixed_base_mul) + ark355_msm_dev with scalars around the negation
    threshold of MsmPlan::negate_high -- 0, 1, r - 1, (r - 1) / 2, (r + 1) / 2, 2^(bits - 1) and random ones -- against
    (sum k_i s_i) * G.  Any size: no naive MSM on the Python side."""
    rnd = random.Random(seed + n)
    G_ = g1(C) if group == 1 else g2(C)
    raw = Z.g1_raw if group == 1 else Z.g2_raw
    fromraw = Z.g1_from_raw if group == 1 else Z.g2_from_raw
    sz = lib.sizes(C.curve_id)
    psz = sz["g1"] if group == 1 else sz["g2"]
    ss = [rnd.getrandbits(64) + 1 for _ in range(n)]
    bases = lib.fixed_base_mul(ctx, C.curve_id, group, raw(C, G_.gen), b"".join(Z.fr_canon(C, s) for s in ss), n, psz)
    half = (C.r - 1) // 2
    special = [0, 1, C.r - 1, half, half + 1, half - 1, 1 << (C.r.bit_length() - 1), C.r - 2, 2, half + 12345]
    ks = [special[i] if i < len(special) else rnd.randrange(C.r) for i in range(n)]
    h = lib.bases_load(ctx, C.curve_id, group, bases, n)
    try:
        ptr, keep = to_dev(b"".join(Z.fr_canon(C, k) for k in ks))
        out = lib.msm_dev(ctx, h, ptr, n, 0, psz)
        expect = G_.mul(G_.gen, sum(k * s for k, s in zip(ks, ss)) % C.r)
        assert fromraw(C, out) == expect, (C.name, group, n)
    finally:
        lib.dll.ark355_bases_free(h)
