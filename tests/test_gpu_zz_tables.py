"""GPU tier, last file of the run on purpose (new in round 3): window tables with a stride (the fallback for keys whose
full tables would not fit HBM, MsmPlan::wstride) and the planner behind ark355_pk_load, on the device -- the same cases
tests/test_emul_o3.py runs on the emulator.  At BASELINE size the stride-2 layout was measured through bench.py
(ARK355_TABLE_STRIDE=2, every proof == closed form; profiles/r03_epilogue_ab.txt)."""
import numpy as np
import pytest

import o3_cases as O
from oracle import synthetic as S
from oracle.fields import BLS12_381, BN254

pytestmark = pytest.mark.gpu


def _to_dev(b):
    import torch
    d = torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    return d.data_ptr(), d


@pytest.mark.parametrize("stride", ["2", "3", "16"])
def test_strided_window_tables(gpu_lib, gpu_ctx, gpu_policy, stride):
    gpu_policy.setenv("ARK355_TABLE_STRIDE", stride)
    O.check_resident_msm(gpu_lib, gpu_ctx, BLS12_381, 1, 5000, _to_dev, seed=21)      # c = 8: 32 windows
    O.check_resident_msm(gpu_lib, gpu_ctx, BN254, 2, 700, _to_dev, seed=22)           # c = 4: 64 windows
    if stride != "16":
        O.check_instance(gpu_lib, gpu_ctx, BLS12_381, S.mulchain_csr(BLS12_381.r, 3000), [(5, 7)])


def test_table_budget_picks_a_stride_or_reports_enomem(gpu_lib, gpu_ctx, gpu_policy):
    from oracle.c import cbase
    from snark_amd._binding import Ark355Error, ENOMEM
    C = BLS12_381
    inst = S.mulchain_csr(C.r, 20000)
    n, ell, w, mats, z = inst
    pk, _ = cbase.setup_raw_c(C, n, ell, w, mats, O.TD)
    pkh, rh = O.load(gpu_lib, gpu_ctx, C, inst, pk)
    full = gpu_lib.pk_table_info(pkh)
    O.free(gpu_lib, pkh, rh)
    assert full["table_stride"] == 1
    # a budget the unpacked BLS12-381 rows (128 B) miss and the packed ones (96 B) meet: the planner packs the rows before it
    # gives up windows (round 5) -- stride 1, three quarters of the bytes, same proof
    gpu_policy.setenv("ARK355_HBM_BUDGET_MB", str(max(1, (full["table_bytes"] * 85 // 100) >> 20)))
    pkh, rh = O.load(gpu_lib, gpu_ctx, C, inst, pk)
    try:
        info = gpu_lib.pk_table_info(pkh)
        assert info["table_stride"] == 1 and info["table_bytes"] * 100 < full["table_bytes"] * 80, (info, full)
        zb = S._mont_bytes(C.r, z)
        got = gpu_lib.prove(gpu_ctx, pkh, rh, zb, len(z), O.Z.fr_canon(C, 3), O.Z.fr_canon(C, 4), gpu_lib.sizes(C.curve_id))
        assert got == cbase.prove(C, n, ell, w, mats, zb, pk, 3, 4)
    finally:
        O.free(gpu_lib, pkh, rh)
    gpu_policy.setenv("ARK355_HBM_BUDGET_MB", str(max(1, (full["table_bytes"] // 3) >> 20)))
    pkh, rh = O.load(gpu_lib, gpu_ctx, C, inst, pk)
    try:
        info = gpu_lib.pk_table_info(pkh)
        assert info["table_stride"] > 1 and info["table_bytes"] < full["table_bytes"] // 2
        zb = S._mont_bytes(C.r, z)
        got = gpu_lib.prove(gpu_ctx, pkh, rh, zb, len(z), O.Z.fr_canon(C, 3), O.Z.fr_canon(C, 4), gpu_lib.sizes(C.curve_id))
        assert got == cbase.prove(C, n, ell, w, mats, zb, pk, 3, 4)
    finally:
        O.free(gpu_lib, pkh, rh)
    gpu_policy.setenv("ARK355_HBM_BUDGET_MB", "1")
    with pytest.raises(Ark355Error) as e:
        O.load(gpu_lib, gpu_ctx, C, inst, pk)
    assert e.value.code == ENOMEM and "do not fit" in str(e.value)
