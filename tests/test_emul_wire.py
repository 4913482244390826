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
