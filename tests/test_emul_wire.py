"""Wire formats behind the C ABI on the CPU emulator build (see tests/wire_cases.py)."""
import pytest

import wire_cases as W
from oracle.fields import BLS12_381, BN254


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_point_codecs(emul_lib, emul_ctx, C):
    W.points_case(emul_lib, emul_ctx, C)


@pytest.mark.parametrize("C,compressed", [(BLS12_381, False), (BLS12_381, True), (BN254, True)],
                         ids=["bls-uncompressed", "bls-compressed", "bn-compressed"])
def test_key_stream_to_proof_bytes(emul_lib, emul_ctx, C, compressed):
    W.key_stream_case(emul_lib, emul_ctx, C, n=12, compressed=compressed)


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_validation_modes(emul_lib, emul_ctx, C):
    W.validation_case(emul_lib, emul_ctx, C)


@pytest.mark.parametrize("compressed", [False, True])
def test_c_oracle_key_stream_loads_and_proves(emul_lib, emul_ctx, compressed):
    """The stream builder of the BASELINE-size GPU test (tests/test_gpu_wire_large.py: raw key of the oracle's C generator ->
    cbase.pk_stream) at a size the emulator handles: ark355_pk_load_bytes -> ark355_prove == cbase.prove on the raw key."""
    import o3_cases as O
    from oracle import serialize as Z, synthetic as S
    from oracle.c import cbase
    C = BLS12_381
    inst = S.mulchain_csr(C.r, 20)
    n, ell, w, mats, z = inst
    pk = O.oracle_key(C, inst)
    stream = cbase.pk_stream(C, pk, compressed)
    pkh = emul_lib.pk_load_bytes(emul_ctx, C.curve_id, stream, compressed=compressed, validate=1)
    rh = emul_lib.r1cs_load(emul_ctx, C.curve_id, n, ell, w, mats)
    try:
        zb = S._mont_bytes(C.r, z)
        got = emul_lib.prove(emul_ctx, pkh, rh, zb, len(z), Z.fr_canon(C, 5), Z.fr_canon(C, 6), emul_lib.sizes(C.curve_id))
        assert got == O.oracle_prove(C, inst, zb, pk, 5, 6)
    finally:
        emul_lib.dll.ark355_pk_free(pkh)
        emul_lib.dll.ark355_r1cs_free(rh)
