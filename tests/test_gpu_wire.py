"""Wire formats behind the C ABI on the GPU (see tests/wire_cases.py): device point codecs and the end-to-end path
ProvingKey bytes -> ark355_pk_load_bytes -> ark355_prove -> proof bytes == the oracle's."""
import pytest

import wire_cases as W
from oracle.fields import BLS12_381, BN254

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_point_codecs(gpu_lib, gpu_ctx, C):
    W.points_case(gpu_lib, gpu_ctx, C, n=40)


@pytest.mark.parametrize("C,compressed", [(BLS12_381, False), (BLS12_381, True), (BN254, False), (BN254, True)],
                         ids=["bls-uncompressed", "bls-compressed", "bn-uncompressed", "bn-compressed"])
def test_key_stream_to_proof_bytes(gpu_lib, gpu_ctx, C, compressed):
    W.key_stream_case(gpu_lib, gpu_ctx, C, n=150, compressed=compressed)


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_validation_modes(gpu_lib, gpu_ctx, C):
    """Validate::Yes semantics on the device decoders and on the proof path: subgroup membership, canonical infinity,
    flag combinations (tests/wire_cases.py)."""
    W.validation_case(gpu_lib, gpu_ctx, C)
