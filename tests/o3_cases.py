"""Checks of the C ABI against the independent C oracle (O3: oracle/c), shared by the GPU tier at BASELINE sizes
(tests/test_gpu_o3_large.py) and the CPU-emulator tier at tiny sizes (tests/test_emul_o3.py).  The proving key comes
from the ORACLE's generator (cbase.setup_raw_c), never from the product's setup."""
import random

import numpy as np

from oracle import groth16 as G, serialize as Z, synthetic as S
from oracle.c import cbase

TD = G.Trapdoor(tau=0x1F3D5B79A8C6E4F2_0123456789ABCDEF, alpha=0xA1, beta=0xB2B2, gamma=0xC3C3C3, delta=0xD4D4D4D4)


# Oracle keys and oracle proofs are the expensive part of the BASELINE-size tests (2^20: generator ~10 s, a CPU proof ~7 s on
# 16 cores): one generator run per (curve, instance) and one CPU proof per (curve, instance, r, s) for the whole session.
_KEYS = {}
_PROOFS = {}
_KEYS_MAX = 6          # a 2^20 BLS12-381 key is ~0.6 GB of host memory


def oracle_key(C, inst):
    n, ell, w, mats, z = inst
    k = (C.name, n, ell, w)
    if k not in _KEYS:
        while len(_KEYS) >= _KEYS_MAX:
            old = next(iter(_KEYS))
            _KEYS.pop(old)
            for pk_ in [q for q in _PROOFS if q[:4] == old]:
                _PROOFS.pop(pk_)
        _KEYS[k] = cbase.setup_raw_c(C, n, ell, w, mats, TD)[0]
    return _KEYS[k]


def oracle_prove(C, inst, zb, pk, r_, s_):
    n, ell, w, mats, z = inst
    k = (C.name, n, ell, w, r_, s_)
    if k not in _PROOFS:
        _PROOFS[k] = cbase.prove(C, n, ell, w, mats, zb, pk, r_, s_)
    return _PROOFS[k]


def load(lib, ctx, C, inst, pk):
    n, ell, w, mats, z = inst
    N = 1
    while N < n + ell:
        N <<= 1
    pkh = lib.pk_load(ctx, C.curve_id, ell, w, N, pk["a_query"], pk["b_g1_query"], pk["b_g2_query"], pk["h_query"],
                      pk["l_query"], pk["alpha_g1"], pk["beta_g1"], pk["delta_g1"], pk["beta_g2"], pk["delta_g2"])
    rh = lib.r1cs_load(ctx, C.curve_id, n, ell, w, mats)
    return pkh, rh


def free(lib, pkh, rh):
    lib.dll.ark355_pk_free(pkh)
    lib.dll.ark355_r1cs_free(rh)


def vk_parts(pk):
    return (pk["alpha_g1"], pk["beta_g2"], pk["gamma_g2"], pk["delta_g2"], pk["gamma_abc_g1"])


def check_equation(lib, ctx, C, inst, pk, proofs, python_pairing=False):
    """The Groth16 equation on proofs made with the ORACLE's key: ark355_verify_batch (device MSMs + the library's host
    pairing) and, once per call when asked, the oracle's independent textbook pairing (oracle/pairing.py)."""
    n, ell, w, mats, z = inst
    inputs = S._mont_bytes(C.r, z[1:ell])
    rnd = random.Random(len(proofs) * 977 + n)
    rho = [Z.fr_canon(C, rnd.randrange(1, C.r)) for _ in proofs]
    assert lib.verify_batch(ctx, C.curve_id, vk_parts(pk), proofs, inputs * len(proofs), rho if len(proofs) > 1 else None)
    # a proof of the same statement with C replaced by A must fail
    a, b, c = proofs[0]
    assert not lib.verify_batch(ctx, C.curve_id, vk_parts(pk), [(a, b, a)], inputs)
    if python_pairing:
        vk = G.VerifyingKey(Z.g1_from_raw(C, pk["alpha_g1"]), Z.g2_from_raw(C, pk["beta_g2"]), Z.g2_from_raw(C, pk["gamma_g2"]),
                            Z.g2_from_raw(C, pk["delta_g2"]),
                            [Z.g1_from_raw(C, pk["gamma_abc_g1"][i * len(a):(i + 1) * len(a)]) for i in range(ell)])
        assert G.verify(C, vk, list(z[1:ell]), G.Proof(Z.g1_from_raw(C, a), Z.g2_from_raw(C, b), Z.g1_from_raw(C, c)))


def check_instance(lib, ctx, C, inst, rs_pairs, batch=0, inflight=4, sharded=False, equation=True, python_pairing=False,
                   timing=None, self_exchange=True):
    """Key from the oracle's generator; `ark355_prove` (and, with `sharded`, `ark355_prove_sharded` over the real RCCL at
    world size 1 with both exchange modes; with `batch`, `ark355_prove_batch`) byte-compared with `cbase.prove`; with
    `equation` every proof is also put through the Groth16 equation (check_equation)."""
    import time
    n, ell, w, mats, z = inst
    t0 = time.perf_counter()
    pk = oracle_key(C, inst)
    t1 = time.perf_counter()
    zb = S._mont_bytes(C.r, z)
    sizes = lib.sizes(C.curve_id)
    pkh, rh = load(lib, ctx, C, inst, pk)
    t2 = time.perf_counter()
    comm = None
    try:
        assert lib.is_satisfied(ctx, rh, zb, len(z)) == -1
        proofs = []
        for r_, s_ in rs_pairs:
            got = lib.prove(ctx, pkh, rh, zb, len(z), Z.fr_canon(C, r_), Z.fr_canon(C, s_), sizes)
            exp = oracle_prove(C, inst, zb, pk, r_, s_)
            assert got == exp, (C.name, n, "ark355_prove vs oracle/c")
            proofs.append(got)
            if sharded:
                # a whole key is shard 0 of 1: communicator creation, the all-gather / the (empty) ring and the combine
                # run exactly as on 8 GPUs
                from snark_amd._binding import SHARD_BUCKET_RING, SHARD_WINDOW
                if comm is None:
                    comm = lib.comm_init(ctx, lib.comm_unique_id(), 0, 1)
                for mode in (SHARD_WINDOW, SHARD_BUCKET_RING):
                    got_s = lib.prove_sharded(ctx, comm, pkh, rh, zb, len(z), Z.fr_canon(C, r_), Z.fr_canon(C, s_), sizes,
                                              mode=mode)
                    assert got_s == exp, (C.name, n, "ark355_prove_sharded vs oracle/c", mode)
        if sharded and self_exchange:
            # Policy RCCL_SELF: the key loaded AGAIN, now in the layout of the distributed witness map, whose three all-to-all
            # exchanges run as grouped ncclSend / ncclRecv pairs of rank 0 with itself, and the bucket ring makes one step with
            # itself per MSM -- the point-to-point calls of an 8-GPU proof on the one GPU there is, compared with the ORACLE's
            # proof at this size (the self exchange had only met the package's own closed form at n = 300).
            # The policy is reset BEFORE the proofs: the key, not the policy of the moment, says that the rank is its own peer.
            lib.ctx_set_policy(ctx, "RCCL_SELF", 1)
            try:
                pkh2, rh2 = load(lib, ctx, C, inst, pk)
            finally:
                lib.ctx_set_policy(ctx, "RCCL_SELF", 0)
            try:
                r_, s_ = rs_pairs[0]
                exp = oracle_prove(C, inst, zb, pk, r_, s_)
                got_s = lib.prove_sharded(ctx, comm, pkh2, rh2, zb, len(z), Z.fr_canon(C, r_), Z.fr_canon(C, s_), sizes, mode=SHARD_WINDOW)
                assert got_s == exp, (C.name, n, "RCCL_SELF: distributed witness map over ncclSend / ncclRecv vs oracle/c")
                lib.ctx_set_policy(ctx, "RCCL_SELF", 1)
                got_s = lib.prove_sharded(ctx, comm, pkh2, rh2, zb, len(z), Z.fr_canon(C, r_), Z.fr_canon(C, s_), sizes, mode=SHARD_BUCKET_RING)
                assert got_s == exp, (C.name, n, "RCCL_SELF: bucket ring with itself vs oracle/c")
                lib.ctx_set_policy(ctx, "RCCL_SELF", 0)
                # the same key through the plain entry point: replicated map + gather of the rank's coefficients
                assert lib.prove(ctx, pkh2, rh2, zb, len(z), Z.fr_canon(C, r_), Z.fr_canon(C, s_), sizes) == exp
            finally:
                lib.ctx_set_policy(ctx, "RCCL_SELF", 0)
                free(lib, pkh2, rh2)
        t3 = time.perf_counter()
        if batch:
            rnd = random.Random(batch)
            rs = [(rnd.randrange(C.r), rnd.randrange(C.r)) for _ in range(batch)]
            outs = lib.prove_batch(ctx, pkh, rh, [zb] * batch, len(z), [Z.fr_canon(C, a) for a, _ in rs],
                                   [Z.fr_canon(C, b) for _, b in rs], sizes, inflight=inflight)
            for (r_, s_), got in zip(rs, outs):
                assert got == cbase.prove(C, n, ell, w, mats, zb, pk, r_, s_), (C.name, n, "ark355_prove_batch")
            proofs += outs
        if equation and proofs:
            check_equation(lib, ctx, C, inst, pk, proofs, python_pairing=python_pairing)
        if timing is not None:
            timing.update(oracle_setup_s=t1 - t0, key_load_s=t2 - t1, prove_and_oracle_s=t3 - t2)
    finally:
        if comm is not None:
            lib.comm_destroy(comm)
        free(lib, pkh, rh)


def scalars(C, n, dist, seed):
    """MSM micro-benchmark scalar distributions of SURVEY.md 8d, canonical 32-byte LE."""
    rnd = random.Random(seed)
    top = (1 << (C.r.bit_length() - 1 - 192)) - 1                        # values < 2^(bits-1) < r: canonical
    if dist == "uniform":
        raw = np.frombuffer(rnd.randbytes(32 * n), dtype="<u8").reshape(n, 4).copy()
        raw[:, 3] &= top
        return raw.tobytes()
    if dist == "equal":
        return Z.fr_canon(C, rnd.randrange(C.r)) * n
    assert dist == "boolean"                                             # 90 % in {0, 1}, 10 % uniform
    raw = np.zeros((n, 4), dtype="<u8")
    kind = np.frombuffer(rnd.randbytes(n), dtype=np.uint8)
    raw[:, 0] = (kind & 1)
    uni = np.frombuffer(rnd.randbytes(32 * n), dtype="<u8").reshape(n, 4).copy()
    uni[:, 3] &= top
    sel = kind >= 230
    raw[sel] = uni[sel]
    return raw.tobytes()


def bases(C, group, n):
    """P_i = (i + 1) * G  (SURVEY.md 8d MSM micro-benchmark), from the oracle's fixed-base routine."""
    ks = np.zeros((n, 4), dtype="<u8")
    ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    gen = Z.g1_raw(C, C.g1_gen) if group == 1 else Z.g2_raw(C, C.g2_gen)
    return cbase.fixed_base(C, group, gen, ks.tobytes(), n)


def check_resident_msm(lib, ctx, C, group, n, to_dev, seed=1, dists=("uniform", "equal", "boolean")):
    """ark355_bases_load + ark355_msm_dev (window tables, radix-2^28 accumulation) vs cbase.msm for the three
    distributions.  to_dev(bytes) -> (device pointer, keepalive)."""
    sz = lib.sizes(C.curve_id)
    psz = sz["g1"] if group == 1 else sz["g2"]
    pts = bases(C, group, n)
    bh = lib.bases_load(ctx, C.curve_id, group, pts, n)
    try:
        for dist in dists:
            sc = scalars(C, n, dist, seed=seed * 10 + group)
            ptr, keep = to_dev(sc)
            got = lib.msm_dev(ctx, bh, ptr, n, 0, psz)
            assert got == cbase.msm(C, group, pts, sc, n), (group, n, dist)
    finally:
        lib.dll.ark355_bases_free(bh)


def check_ntt_full(lib, ctx, C, log_n, seed=5):
    """ark355_ntt_fr vs cb_ntt, four modes, whole vectors (Montgomery images in and out on both sides)."""
    n = 1 << log_n
    rng = np.random.default_rng(seed * 100 + log_n)
    raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << (C.r.bit_length() - 1 - 192)) - 1)      # < 2^(bits-1) < r: valid residues
    raw = raw.astype("<u8")
    data = raw.tobytes()
    for inv, cos in ((0, 0), (1, 0), (0, 1), (1, 1)):
        got = lib.ntt(ctx, C.curve_id, data, log_n, inv, cos)
        exp = cbase.ntt(C, data, log_n, bool(inv), bool(cos))
        assert got == exp, (C.name, log_n, inv, cos)


def check_witness_map_full(lib, ctx, C, inst, dist_worlds=()):
    """dist_worlds: also the distributed map (witness_dist_impl.cuh; all ranks on this device, ark355_witness_map_dist_sim)
    for these world sizes, every coefficient against the same oracle vector."""
    n, ell, w, mats, z = inst
    zb = S._mont_bytes(C.r, z)
    rh = lib.r1cs_load(ctx, C.curve_id, n, ell, w, mats)
    try:
        got = lib.witness_map(ctx, rh, zb, len(z), 32)
        exp = cbase.witness_map(C, n, ell, w, mats, zb)
        assert len(got) == len(exp) and got == exp, (C.name, n)
        # an assignment that does NOT satisfy the constraints: the reference's map has no satisfaction check and returns
        # coset_ifft((a'b' - c') / Z(g)) whatever z is; the six-transform form (witness_impl.cuh) must agree there too
        zbad = list(z)
        zbad[len(z) // 2] = (zbad[len(z) // 2] + 12345) % C.r
        zbb = S._mont_bytes(C.r, zbad)
        assert lib.witness_map(ctx, rh, zbb, len(z), 32) == cbase.witness_map(C, n, ell, w, mats, zbb), (C.name, n, "unsatisfied")
        for world in dist_worlds:
            got = lib.witness_map_dist_sim(ctx, rh, zb, len(z), 32, world)
            assert got == exp, (C.name, n, "distributed over", world)
    finally:
        lib.dll.ark355_r1cs_free(rh)
