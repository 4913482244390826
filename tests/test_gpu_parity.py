"""GPU tier (-m gpu): the HIP path through the C ABI against the oracle -- bit-exact.

Small sizes: full comparison with the oracle's naive algorithms.  Large sizes (up to BASELINE.json's
2^20): size-independent properties -- NTT round trips and DC/Nyquist bins, MSM over bases with known
discrete logs, proofs against the trapdoor closed form and the pairing equation."""
import random

import pytest

import parity_cases as pc
from helpers import r1cs_load_from_rows, z_bytes
from oracle import groth16 as G, serialize as Z, synthetic as S
from oracle.fields import BLS12_381, BN254

pytestmark = pytest.mark.gpu
CURVES = [BLS12_381, BN254]


def test_native_library_is_the_one_loaded(gpu_lib):
    import snark_amd
    assert gpu_lib.path == snark_amd.LIB_PATH and gpu_lib.path.endswith("libark355.so")
    maps = open("/proc/self/maps").read()
    assert "libark355.so" in maps and "libark355_emul" not in maps


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [1, 2, 3, 7, 8, 9, 10, 11, 13])
def test_ntt_vs_oracle(gpu_lib, gpu_ctx, C, log_n):
    pc.ntt_case(gpu_lib, gpu_ctx, C, log_n)


@pytest.mark.parametrize("C,log_n", [(BLS12_381, 16), (BLS12_381, 17), (BLS12_381, 21), (BN254, 20)],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_ntt_roundtrip_large(gpu_lib, gpu_ctx, C, log_n):
    pc.ntt_roundtrip_case(gpu_lib, gpu_ctx, C, log_n)


def test_ntt_two_adicity_error(gpu_lib, gpu_ctx):
    with pytest.raises(Exception) as e:
        gpu_lib.ntt(gpu_ctx, BN254.curve_id, bytes(32), 41, 0, 0)
    assert e.value.code == -18


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("n", [0, 1, 2, 31, 33, 257, 1500])
def test_msm_vs_naive(gpu_lib, gpu_ctx, C, group, n):
    if group == 2 and n > 300:
        n = 300
    pc.msm_case(gpu_lib, gpu_ctx, C, group, n)


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group,n", [(1, 24), (1, 700), (2, 24), (2, 130)])
def test_resident_tables_exceptional_additions(gpu_lib, gpu_ctx, C, group, n):
    """bases_load + msm_dev: the radix-2^28 (G1) / lane-split (G2) bucket kernels over window tables, with
    P + P, P + (-P), infinity and repeated points inside buckets."""
    import numpy as np
    import torch

    def to_dev(b):
        t = torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).cuda()
        return t.data_ptr(), t
    pc.resident_msm_edge_case(gpu_lib, gpu_ctx, C, group, n, to_dev)


@pytest.mark.parametrize("pack", ["1", "0"], ids=["packed", "unpacked"])
def test_resident_tables_both_row_formats(gpu_lib, gpu_ctx, gpu_policy, pack):
    """Policy PACK_ROWS: bit-packed 28-bit table rows (parked-flush kernels) and one word per limb (plain walk), each forced on
    both curves and groups: exceptional additions inside buckets against the oracle's naive MSM, a 2^16-term MSM with the three
    scalar distributions against the C oracle, a whole proof."""
    import numpy as np
    import torch
    import o3_cases as O
    gpu_policy.setenv("ARK355_PACK_ROWS", pack)

    def to_dev(b):
        t = torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).cuda()
        torch.cuda.synchronize()
        return t.data_ptr(), t
    for C in CURVES:
        for group in (1, 2):
            pc.resident_msm_edge_case(gpu_lib, gpu_ctx, C, group, 200, to_dev)
        A, B, Cm, z, ell = S.mulchain_direct(C.r, 300)
        pc.prove_case(gpu_lib, gpu_ctx, C, A, B, Cm, z, ell, rs=((0x1234567890abcdef, 0xfedcba0987654321aabbccdd),))
    for group in (1, 2):
        O.check_resident_msm(gpu_lib, gpu_ctx, BLS12_381, group, 1 << 16, to_dev, seed=3 + int(pack))


@pytest.mark.parametrize("C,group,n,skew", [
    (BLS12_381, 1, 1 << 14, None), (BLS12_381, 1, 1 << 14, "equal"), (BLS12_381, 1, 1 << 14, "boolean"),
    (BLS12_381, 2, 1 << 12, None), (BN254, 1, 1 << 14, None), (BN254, 2, 1 << 12, "boolean"),
    (BLS12_381, 1, 1 << 18, None), (BLS12_381, 1, (1 << 16) + 77, "boolean"),
], ids=lambda v: getattr(v, "name", str(v)))
def test_msm_known_dlog(gpu_lib, gpu_ctx, C, group, n, skew):
    pc.msm_known_dlog_case(gpu_lib, gpu_ctx, C, group, n, skew=skew)


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_r1cs_ops_and_witness_map(gpu_lib, gpu_ctx, C):
    for cs in (S.bench_lc_cs(C.r, 300), S.dummy_cs(C.r, 128), S.mulchain_cs(C.r, 1), S.mulchain_cs(C.r, 255)):
        A, B, Cm, z, ell = S.cs_to_instance(cs)
        pc.r1cs_case(gpu_lib, gpu_ctx, C, A, B, Cm, z, ell)


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_prove_small_vs_oracle_and_pairing(gpu_lib, gpu_ctx, C):
    A, B, Cm, z, ell = S.mulchain_direct(C.r, 13)
    pc.prove_case(gpu_lib, gpu_ctx, C, A, B, Cm, z, ell, verify=True,
                  rs=((0x1234567890abcdef, 0xfedcba0987654321aabbccdd), (0, 5), (7, 0), (C.r - 1, C.r - 2)))
    A, B, Cm, z, ell = S.cs_to_instance(S.bench_lc_cs(C.r, 100))
    pc.prove_case(gpu_lib, gpu_ctx, C, A, B, Cm, z, ell)


def test_prove_dummy_circuit_config1(gpu_lib, gpu_ctx):
    """BASELINE.json configs[0]: the reference's DummyCircuit shape (sr1cs/mod.rs:276-319) at 2^10: all scalars
    equal -- the worst case for bucket collisions."""
    C = BLS12_381
    A, B, Cm, z, ell = S.cs_to_instance(S.dummy_cs(C.r, 1 << 10))
    pc.prove_case(gpu_lib, gpu_ctx, C, A, B, Cm, z, ell, verify=True)


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_prove_reference_example_circuit_config0(gpu_lib, gpu_ctx, C):
    """BASELINE.json configs[0]: the reference's own example program (relations/examples/satisfiable.rs:7-150, inputs
    3, 4, 6, 7 -> 198) proven on the GPU, byte-equal to the oracle and accepted by its pairing check; the
    non_satisfiable.rs witness is reported at constraint 1 by the device check and still proves nothing valid."""
    from oracle import r1cs as R
    cs = R.ConstraintSystem(C.r)
    R.example_circuit(cs, satisfiable=True)
    A, B, Cm, z, ell = S.cs_to_instance(cs)
    pc.prove_case(gpu_lib, gpu_ctx, C, A, B, Cm, z, ell, verify=True)
    bad = R.ConstraintSystem(C.r)
    R.example_circuit(bad, satisfiable=False)
    A2, B2, C2, z2, _ = S.cs_to_instance(bad)
    rh = r1cs_load_from_rows(gpu_lib, gpu_ctx, C, A2, B2, C2, ell, len(z2) - ell)
    try:
        assert gpu_lib.is_satisfied(gpu_ctx, rh, z_bytes(C, z2), len(z2)) == 1
        assert gpu_lib.is_satisfied(gpu_ctx, rh, z_bytes(C, z), len(z)) == -1
    finally:
        gpu_lib.dll.ark355_r1cs_free(rh)


def _host_mirror_case(curve_name, n, seed):
    """Product setup (host scalars + device fixed-base) + prove vs the closed form; oracle re-derives the
    closed-form exponents independently from the same trapdoor and checks the pairing equation."""
    from snark_amd import params, synthetic
    from snark_amd.groth16 import Groth16
    cv = params.CURVES[curve_name]
    C = BLS12_381 if curve_name == "bls12_381" else BN254
    r1, z = synthetic.mulchain(cv, n, seed=seed)
    g = Groth16(cv)
    try:
        rnd = random.Random(seed)
        pk, vk = g.circuit_specific_setup(r1, lambda: rnd.randrange(1, C.r), keep_trapdoor=True)
        assert g.is_satisfied(r1, z) is None
        r_, s_ = rnd.randrange(C.r), rnd.randrange(C.r)
        proof = g.prove(pk, r1, synthetic.z_to_mont_bytes(cv, z), r=r_, s=s_)
        assert proof == g.prove_closed_form(pk, z, r_, s_)
        # independent re-derivation by the oracle from the same toxic waste
        td = pk.trapdoor
        A, B, Cm, z2, ell = S.mulchain_direct(C.r, n, seed=seed)
        assert z2 == z
        u, v, w, zt, dom = G.qap_scalars(C, A, B, Cm, n, ell, len(z), td["tau"])
        assert (u, v, w) == (td["u"], td["v"], td["w"])
        opk = G.ProvingKey(vk=None, beta_g1=None, delta_g1=None, a_query=[], b_g1_query=[], b_g2_query=[],
                           h_query=[], l_query=[],
                           trapdoor=G.Trapdoor(td["tau"], td["alpha"], td["beta"], td["gamma"], td["delta"]),
                           u=u, v=v, w=w)
        exp = G.prove_closed_form(C, opk, z, ell, r_, s_)
        got = G.Proof(Z.g1_from_raw(C, proof.a), Z.g2_from_raw(C, proof.b), Z.g1_from_raw(C, proof.c))
        assert got == exp
        s1 = g.sizes["g1"]
        ovk = G.VerifyingKey(alpha_g1=Z.g1_from_raw(C, vk.alpha_g1), beta_g2=Z.g2_from_raw(C, vk.beta_g2),
                             gamma_g2=Z.g2_from_raw(C, vk.gamma_g2), delta_g2=Z.g2_from_raw(C, vk.delta_g2),
                             gamma_abc_g1=[Z.g1_from_raw(C, vk.gamma_abc_g1[i * s1:(i + 1) * s1]) for i in range(ell)])
        assert G.verify(C, ovk, z[1:ell], got)
        # a wrong witness gives a different proof that must NOT verify
        zb = list(z)
        zb[5] = (zb[5] + 1) % C.r
        from oracle import r1cs as R
        assert g.is_satisfied(r1, zb) == R.first_unsatisfied_r1cs(A, B, Cm, zb, C.r) == 1
        bad = g.prove(pk, r1, synthetic.z_to_mont_bytes(cv, zb), r=r_, s=s_)
        badp = G.Proof(Z.g1_from_raw(C, bad.a), Z.g2_from_raw(C, bad.b), Z.g1_from_raw(C, bad.c))
        assert not G.verify(C, ovk, z[1:ell], badp)
    finally:
        g.close()


@pytest.mark.parametrize("curve_name,n", [("bls12_381", 1 << 12), ("bn254", 1 << 12), ("bls12_381", (1 << 16) - 100)])
def test_host_mirror_setup_prove_closed_form(gpu_lib, curve_name, n):
    _host_mirror_case(curve_name, n, seed=0x355)


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_prove_batch_in_flight(gpu_lib, gpu_ctx, C):
    """ark355_prove_batch: 7 proofs, 3 in flight on private streams, each byte-checked against the closed form;
    then a larger circuit with 4 in flight."""
    pc.prove_batch_case(gpu_lib, gpu_ctx, C, count=7, n=40, inflight=3)
    pc.prove_batch_case(gpu_lib, gpu_ctx, C, count=4, n=500, inflight=4)


def test_sharded_prove_over_rccl():
    """snark_amd.parallel.ShardedGroth16 over the RCCL backend (world_size 1 on a one-GPU box): partial sums gathered
    from HBM with all_gather_into_tensor, combined by ark355_prove_combine, equal to the whole-key proof and to the
    closed form.  Runs in its own process so that the process group cannot leak into other tests."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "rccl_single_rank.py")], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rccl_ok 1" in out.stdout
    # the second pass of the helper: the rank as its own RCCL peer (grouped ncclSend / ncclRecv to self in the witness map's
    # all-to-alls and in the bucket ring), byte-identical proofs
    assert "rccl_self 1 ok" in out.stdout


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("group", [1, 2])
def test_fixed_base_mul_vs_oracle(gpu_lib, gpu_ctx, C, group):
    pc.fixed_base_case(gpu_lib, gpu_ctx, C, group, n=4000)


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_batch_verification_vs_oracle_pairing(gpu_lib, gpu_ctx, C):
    pc.verify_batch_case(gpu_lib, gpu_ctx, C, count=6)


@pytest.mark.gpu
def test_prove_with_window_17_negating_high_scalars(gpu_lib, gpu_ctx, gpu_policy):
    """ARK355_MSM_C=17 on a key of >= 1024 terms: MsmPlan::negate_high (scalars above (r - 1) / 2 become r - k with
    flipped digit signs), 15 windows, 2^16 buckets -- the proof must still be the oracle's."""
    gpu_policy.setenv("ARK355_MSM_C", "17")
    C = BLS12_381
    A, B, Cm, z, ell = S.mulchain_direct(C.r, 1030)
    pc.prove_case(gpu_lib, gpu_ctx, C, A, B, Cm, z, ell, rs=((C.r - 3, 12345),))


@pytest.mark.parametrize("batch,side", [(1, 1), (1, 0), (0, 0), (1, 2)], ids=["batched+side", "batched", "per-msm", "batched+side+g1-aside"])
@pytest.mark.parametrize("circuit", ["mulchain", "dummy"])
def test_one_stream_tail_variants(gpu_lib, gpu_ctx, gpu_policy, circuit, batch, side):
    """A one-stream proof: the four G1 tails as one launch per step (policy BATCH_TAILS) and -- the proof being alone on the
    device -- the G2 tails on a side stream underneath the G1 accumulations (SIDE_G2_TAILS), against the per-MSM tails of
    rounds 1-3; uniform scalars and the all-equal DummyCircuit (heavy buckets).  Proof bytes == oracle every way."""
    gpu_policy.setenv("ARK355_SCHED", "0")
    gpu_policy.setenv("ARK355_BATCH_TAILS", str(batch))
    gpu_policy.setenv("ARK355_SIDE_G2_TAILS", "1" if side else "0")
    gpu_policy.setenv("ARK355_SIDE_G1_TAILS", "1" if side == 2 else "0")     # (round 6) A, B1, L' tails aside as a batch of three
    C = BLS12_381
    inst = S.mulchain_direct(C.r, 700) if circuit == "mulchain" else S.cs_to_instance(S.dummy_cs(C.r, 900))
    pc.prove_case(gpu_lib, gpu_ctx, C, *inst, rs=((11, 0xABCDEF),))


@pytest.mark.parametrize("side", ["1", "0"], ids=["h-tails-on-sort-stream", "h-tails-on-reduction-stream"])
def test_pipeline_last_msm_tails_on_either_stream(gpu_lib, gpu_ctx, gpu_policy, side):
    """Five-stream pipeline (SCHED=1): the tails of the H MSM on the sort stream (policy SIDE_H_TAILS, the default) and behind
    the tails of L' on the reduction stream -- uniform scalars and the all-equal DummyCircuit; proof bytes == oracle both ways."""
    gpu_policy.setenv("ARK355_SCHED", "1")
    gpu_policy.setenv("ARK355_SIDE_H_TAILS", side)
    C = BLS12_381
    pc.prove_case(gpu_lib, gpu_ctx, C, *S.mulchain_direct(C.r, 700), rs=((13, 0xFEDCBA),))
    pc.prove_case(gpu_lib, gpu_ctx, C, *S.cs_to_instance(S.dummy_cs(C.r, 900)), rs=((0, 1),))


@pytest.mark.parametrize("serial", ["1", "0"])
def test_prove_one_stream_schedule(gpu_lib, gpu_ctx, gpu_policy, serial):
    """The schedule prove_run picks with other proofs in flight (one stream) and the one it picks for a proof alone (five
    streams), forced through ARK355_SERIAL: proof bytes == oracle, pairing equation."""
    gpu_policy.setenv("ARK355_SERIAL", serial)
    C = BLS12_381
    A, B, Cm, z, ell = S.mulchain_direct(C.r, 300)
    pc.prove_case(gpu_lib, gpu_ctx, C, A, B, Cm, z, ell, verify=True)


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_check_satisfied_policy(gpu_lib, gpu_ctx, gpu_policy, C):
    """Policy CHECK_SATISFIED (round 6): off, an unsatisfied assignment is proven as the reference's algorithm would prove it
    (oracle O1, seven transforms -- the library's six-transform map agrees for any z); on, ARK355_E_UNSATISFIABLE with the
    index of the first unsatisfied constraint, from ark355_prove and ark355_prove_batch alike."""
    pc.check_satisfied_case(gpu_lib, gpu_ctx, C, gpu_policy, n=300)


def test_box_diagnostics(gpu_lib, gpu_ctx):
    """ark355_diag_mad_rate / ark355_diag_clocks (what bench.py's `roofline.alu.peak` and `box` block are made of): a plausible
    multiply-add rate for a 256-CU part, and counters that move forward at a gfx clock between 0.5 and 3 GHz."""
    import time
    r = gpu_lib.diag_mad_rate(gpu_ctx, 5.0)
    assert 10.0 < r["tmad_per_s"] < 60.0 and 1.0 < r["elapsed_ms"] < 50.0, r
    c0 = gpu_lib.diag_clocks(gpu_ctx)
    gpu_lib.diag_mad_rate(gpu_ctx, 20.0)
    c1 = gpu_lib.diag_clocks(gpu_ctx)
    clk, ms, units = gpu_lib.diag_clocks_delta(c0, c1)
    assert units >= 64, units                         # most of the 256 compute units answered both probes
    assert 15.0 < ms < 500.0, ms                      # the 100 MHz reference: the ~20 ms loop plus two probes
    assert 1000.0 < clk < 2600.0, clk                 # a busy chip runs between 1 and 2.5 GHz
