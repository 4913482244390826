"""BASELINE-size configurations against the independent C oracle (O3), on the GPU, through the C ABI.

Round-1 checked these sizes only product-vs-product (the product's own setup + closed form).  Here the proving key
comes from the ORACLE's generator (`oracle.c.cbase.setup_raw_c`: QAP scalars and fixed-base multiplications by the
C restatement, itself checked against the Python oracle at small sizes in tests/test_oracle_c.py) and the proof bytes
of `ark355_prove` / `ark355_prove_batch` are compared with `cbase.prove` (arkworks-style Pippenger + radix-2 FFT on the
host cores) on the same instance.  This is the window-16 / 64-entry-segment / heavy-merge / two-level-sort regime of
the benchmark (msm_plan, snark_amd/csrc/msm_impl.cuh).

Reference anchors: SNARK::prove (/root/reference/snark/src/lib.rs:50-54); circuits S1 = DummyCircuit
(/root/reference/relations/src/sr1cs/mod.rs:276-319), S3 = bench LCs (/root/reference/relations/examples/bench.rs:22-83),
S2 = SURVEY.md 8d.
"""
import numpy as np
import pytest

import o3_cases as O
from oracle import synthetic as S
from oracle.fields import BLS12_381, BN254

pytestmark = pytest.mark.gpu


def test_s2_2p20_bls12_381_vs_o3(gpu_lib, gpu_ctx):
    """BASELINE configs[1]: S2 mulchain, n = 2^20 (N = 2^21), BLS12-381."""
    C = BLS12_381
    O.check_instance(gpu_lib, gpu_ctx, C, S.mulchain_csr(C.r, 1 << 20), [(0x1234567, 0x89ABCDE)])


def test_s2_2p20_tight_bls12_381_vs_o3(gpu_lib, gpu_ctx):
    """Domain-tight variant n = 2^20 - 100 (N = 2^20)."""
    C = BLS12_381
    O.check_instance(gpu_lib, gpu_ctx, C, S.mulchain_csr(C.r, (1 << 20) - 100), [(C.r - 1, 5)])


def test_s2_2p20_bn254_vs_o3(gpu_lib, gpu_ctx):
    """BASELINE configs[3]: the second field instantiation at n = 2^20."""
    C = BN254
    O.check_instance(gpu_lib, gpu_ctx, C, S.mulchain_csr(C.r, 1 << 20), [(77, C.r - 2)])


def test_batch_2p18_vs_o3(gpu_lib, gpu_ctx):
    """BASELINE configs[4] shape: several 2^18 proofs in flight (ark355_prove_batch), every proof vs oracle/c."""
    C = BLS12_381
    O.check_instance(gpu_lib, gpu_ctx, C, S.mulchain_csr(C.r, 1 << 18), [], batch=4)


def test_s1_dummy_2p18_vs_o3(gpu_lib, gpu_ctx):
    """S1 DummyCircuit at 2^18: every scalar equal (one bucket per window gets everything: heavy-merge path)."""
    C = BLS12_381
    O.check_instance(gpu_lib, gpu_ctx, C, S.dummy_csr(C.r, 1 << 18), [(3, 4)])


def test_s3_bench_lc_2p18_vs_o3(gpu_lib, gpu_ctx):
    """S3 (examples/bench.rs LC shapes, non-unit coefficients, up to 10 terms per LC) at 2^18."""
    C = BLS12_381
    O.check_instance(gpu_lib, gpu_ctx, C, S.bench_lc_csr(C.r, 1 << 18), [(11, 13)])


@pytest.mark.parametrize("group,log_n", [(1, 20), (2, 20), (1, 22)])
def test_resident_msm_vs_o3(gpu_lib, gpu_ctx, group, log_n):
    """ark355_msm_dev over resident window tables at 2^20 / 2^22 vs cbase.msm: uniform, all-equal, boolean scalars."""
    import torch

    def to_dev(b):
        d = torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).cuda()
        torch.cuda.synchronize()
        return d.data_ptr(), d

    O.check_resident_msm(gpu_lib, gpu_ctx, BLS12_381, group, 1 << log_n, to_dev, seed=log_n)
