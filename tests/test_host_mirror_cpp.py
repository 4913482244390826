"""The C++ host mirror of ark-relations / ark-snark (snark_amd/host/*.hpp): its CPU checks mirror the
reference's own unit tests; its GPU path (setup + prove through libark355.so) is compared with the oracle."""
import os
import subprocess

import pytest

from conftest import ROOT

CPP_DIR = os.path.join(ROOT, "tests", "cpp")
EXE = os.path.join(CPP_DIR, "test_host_mirror")


@pytest.fixture(scope="module")
def exe():
    from snark_amd import build
    build.build(verbose=False)
    assert build.build_host_mirror_exe() == EXE
    return EXE


def test_cpp_mirror_cpu_checks(exe):
    """circuit2 golden matrices (gr1cs/tests/mod.rs:136-147), Variable ordering (utils/variable.rs:206-266),
    DummyCircuit (sr1cs/mod.rs:320-330), synthesis-mode quirks."""
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "all CPU checks passed" in out.stdout


def test_cpp_host_synthesis_benchmark(exe):
    """the reference's examples/bench.rs measurement on the C++ mirror (generate_constraints in
    Prove{construct_matrices} mode + finalize).  Checks the shape of the synthesised system; the rate is reported,
    not asserted."""
    n = 1 << 13
    out = subprocess.run([exe, "--synth-bench", "bls12_381", str(n)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    kv = dict(line.split("=", 1) for line in out.stdout.strip().splitlines())
    assert int(kv["synth_constraints"]) == n
    assert int(kv["synth_witnesses"]) == 3 + 3 * n
    assert 3 * n <= int(kv["synth_nnz"]) <= 31 * n
    assert float(kv["synth_constraints_per_s"]) > 0


def _parse(out):
    return {k: bytes.fromhex(v) for k, v in (line.split("=", 1) for line in out.strip().splitlines() if "=" in line)}


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,circuit,n", [("bls12_381", "dummy", 64), ("bls12_381", "mulchain", 100),
                                                  ("bn254", "mulchain", 37), ("bls12_381", "benchlc", 40),
                                                  ("bls12_381", "example", 8), ("bn254", "example", 8)])
def test_cpp_snark_trait_prove_matches_oracle(exe, curve_name, circuit, n):
    from oracle import groth16 as G, serialize as Z, synthetic as S
    from oracle.fields import CURVES
    C = CURVES[curve_name]
    r = subprocess.run([exe, "--prove", curve_name, circuit, str(n)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "batch_ok 1" in r.stdout          # Groth16::prove_batch (ark355_prove_batch) agreed with single proofs
    assert "verify_ok 1" in r.stdout         # Groth16::verify / verify_batch (ark355_verify_batch) on those proofs
    got = _parse(r.stdout)
    if circuit == "example":            # relations/examples/satisfiable.rs, BASELINE configs[0]
        from oracle import r1cs as R
        cs = R.ConstraintSystem(C.r)
        R.example_circuit(cs, satisfiable=True)
        A, B, Cm, z, ell = S.cs_to_instance(cs)
    elif circuit == "dummy":
        A, B, Cm, z, ell = S.cs_to_instance(S.dummy_cs(C.r, n))
    elif circuit == "benchlc":          # S3: random coefficients, repeated columns (same splitmix64 stream in C++)
        A, B, Cm, z, ell = S.cs_to_instance(S.bench_lc_cs(C.r, n))
    else:
        A, B, Cm, z, ell = S.mulchain_direct(C.r, n, start=(0x355, 0x356))
    td = G.Trapdoor(tau=0x1234567, alpha=11, beta=22, gamma=33, delta=44)
    pk = G.setup(C, A, B, Cm, ell, len(z), td)
    for tag, (r_, s_) in (("proof", (0xabcdef01, 0x13579bdf)), ("proof2", (0x777, 0x888))):
        exp = G.prove_closed_form(C, pk, z, ell, r_, s_)
        assert Z.g1_from_raw(C, got[tag + "_a"]) == exp.a
        assert Z.g2_from_raw(C, got[tag + "_b"]) == exp.b
        assert Z.g1_from_raw(C, got[tag + "_c"]) == exp.c
    assert Z.g1_from_raw(C, got["vk_alpha_g1"]) == pk.vk.alpha_g1
    s1 = 2 * C.fq_bytes
    assert [Z.g1_from_raw(C, got["vk_gamma_abc_g1"][i * s1:(i + 1) * s1]) for i in range(ell)] == pk.vk.gamma_abc_g1
    assert G.verify(C, pk.vk, z[1:ell], G.Proof(Z.g1_from_raw(C, got["proof_a"]), Z.g2_from_raw(C, got["proof_b"]),
                                                Z.g1_from_raw(C, got["proof_c"])))


def test_madd28_chains_match_32bit_formulas():
    """tests/cpp/test_madd28_emul.cpp: ~48k G1 and 12k lane-pair G2 mixed additions per curve on radix-2^28 limbs, with
    forced P + P / P + (-P) cases, against the canonical 32-bit formulas; the emulator build traps on any column
    overflow or limb wrap-around on the way."""
    src = os.path.join(CPP_DIR, "test_madd28_emul.cpp")
    exe28 = os.path.join(CPP_DIR, "test_madd28_emul")
    emul = os.path.join(ROOT, "tests", "emul")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-DARK_EMUL", "-w", "-I", emul,
                           "-I", os.path.join(ROOT, "snark_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           src, os.path.join(emul, "hip_emul.cpp"), "-o", exe28, "-lpthread"])
    out = subprocess.run([exe28], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all chains agree" in out.stdout


@pytest.mark.gpu
def test_cpp_end_to_end_pipelined_prove_matches_batch(exe):
    """Groth16::prove_pipelined (synthesis threads feeding in-flight GPU proofs) == ark355_prove_batch proofs with the
    same randomisers; reports end-to-end constraints/s next to the device-only figure."""
    r = subprocess.run([exe, "--e2e", "bls12_381", "3000", "6", "3", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "e2e_ok 1" in r.stdout
    kv = dict(line.split("=", 1) for line in r.stdout.strip().splitlines() if "=" in line)
    assert float(kv["e2e_constraints_per_s"]) > 0 and float(kv["device_only_constraints_per_s"]) > 0


def test_cpp_pipeline_over_the_emulator(emul_lib):
    """The C++ host mirror linked against the EMULATOR build of the same library sources (checker only): setup, then
    Groth16::prove_pipelined (synthesis threads -> the prover), ark355_prove_batch and prove_assignments over page-locked
    buffers must all give the same proofs; --prove additionally checks the proof against the trapdoor closed form.  Covers
    the host-side plumbing (Backend worker contexts, buffer recycling, queueing) on a machine without a GPU."""
    src = os.path.join(CPP_DIR, "test_host_mirror.cpp")
    emul = os.path.join(ROOT, "tests", "emul")
    exe_e = os.path.join(CPP_DIR, "test_host_mirror_emul")
    subprocess.check_call(["g++", "-O2", "-pthread", "-std=c++17", src, "-o", exe_e, "-L" + emul, "-lark355_emul",
                           "-Wl,-rpath," + emul])
    r = subprocess.run([exe_e, "--e2e", "bls12_381", "24", "3", "2", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "e2e_ok 1" in r.stdout
    r = subprocess.run([exe_e, "--prove", "bn254", "mulchain", "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
