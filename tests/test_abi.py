"""The C-ABI library builds for gfx950, loads, and exports every symbol include/ark355.h declares.
No compute call is made here (no GPU in the CPU test tier)."""
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built_lib():
    from snark_amd import build
    path = build.build(verbose=False)
    assert os.path.exists(path)
    return path


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "ark355.h")).read()
    return sorted(set(re.findall(r"\b(ark355_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_match_binding_table():
    from snark_amd import _binding
    assert _declared_symbols() == sorted(_binding.SYMBOLS)


def test_library_loads_and_exports_all_symbols(built_lib):
    import snark_amd
    lib = snark_amd.lib()
    for name in _declared_symbols():
        assert hasattr(lib.dll, name), name
    assert lib.dll.ark355_version() >= 1
    assert lib.sizes(snark_amd.BLS12_381) == {"fr": 32, "fq": 48, "g1": 96, "g2": 192}
    assert lib.sizes(snark_amd.BN254) == {"fr": 32, "fq": 32, "g1": 64, "g2": 128}


def test_no_device_is_a_clean_error(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import snark_amd
    with pytest.raises(snark_amd.Ark355Error) as e:
        snark_amd.lib().ctx_create(0)
    assert e.value.code in (snark_amd.ENODEV, snark_amd.EHIP)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under snark_amd/ may reference it."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "snark_amd")):
        for f in fs:
            if f.endswith((".py", ".h", ".cuh", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "oracle/" in txt and "#include" in txt and "oracle/" in "".join(l for l in txt.splitlines() if l.strip().startswith("#include")):
                    bad.append(f)
    assert not bad, bad


def test_host_only_entry_points_reject_bad_arguments(built_lib):
    """Entry points that need no device (argument checks run before any HIP call): NULL handles, out-of-range `validate`,
    a proof of the wrong length -- ARK355_EINVAL, never a crash."""
    import ctypes as C
    import snark_amd
    lib = snark_amd.lib()
    EINVAL = snark_amd.EINVAL
    c, w, s, b = C.c_uint32(7), C.c_uint32(7), C.c_uint32(7), C.c_uint64(7)
    assert lib.dll.ark355_pk_table_info(None, C.byref(c), C.byref(w), C.byref(s), C.byref(b)) == EINVAL
    assert lib.dll.ark355_pk_dims(None, None, None, None) == EINVAL
    from snark_amd._binding import ProofRaw
    buf = (C.c_uint8 * 512)()
    raw = ProofRaw()
    out = C.byref(raw)
    for validate in (-1, 3):
        assert lib.dll.ark355_proof_from_bytes(snark_amd.BLS12_381, buf, 192, 1, validate, out) == EINVAL
    assert lib.dll.ark355_proof_from_bytes(snark_amd.BLS12_381, buf, 191, 1, 1, out) == EINVAL         # bad length
    assert lib.dll.ark355_proof_from_bytes(99, buf, 192, 1, 1, out) == EINVAL                          # unknown curve
    # an all-zero compressed BLS12-381 "proof": the compressed bit is missing -> bad flags, in every validation mode
    for validate in (0, 1, 2):
        assert lib.dll.ark355_proof_from_bytes(snark_amd.BLS12_381, buf, 192, 1, validate, out) == EINVAL
    # three compressed points at infinity are a well-formed (if useless) proof encoding
    inf = bytes([0xC0]) + bytes(47) + bytes([0xC0]) + bytes(95) + bytes([0xC0]) + bytes(47)
    ib = (C.c_uint8 * 192)(*inf)
    assert lib.dll.ark355_proof_from_bytes(snark_amd.BLS12_381, ib, 192, 1, 1, out) == 0
    assert bytes(raw.a) == bytes(96) and bytes(raw.b) == bytes(192) and bytes(raw.c) == bytes(96)


def test_kernel_statistics_are_read_from_the_built_library(built_lib):
    """tools/code_object_stats.py on the library just built (what snark_amd/build.py writes next to it and bench.py prices its
    integer roofline with): the accumulation kernels of both curves are found in the code objects, their hot paths hold the
    multiply-adds of one mixed addition (thousands, G2 per lane more than G1), the G1 loops make no scratch access and the
    BLS12-381 lane-pair loop -- whose accumulator lives in LDS since round 6 -- only a handful."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import code_object_stats as cos
        st = cos.library_stats(built_lib)
    finally:
        sys.path.remove(os.path.join(ROOT, "tools"))
    for key in ("bls12_381.g1", "bls12_381.g2", "bn254.g1", "bn254.g2"):
        assert key in st, (key, sorted(st))
    g1, g2 = st["bls12_381.g1"]["hot_block"], st["bls12_381.g2"]["hot_block"]
    assert 2500 < g1["multiply_adds"] < 4000 and g1["multiply_adds"] < g2["multiply_adds"] < 6000
    assert st["bn254.g1"]["hot_block"]["multiply_adds"] < g1["multiply_adds"]
    assert max(b["scratch_loads"] for b in (g1, st["bn254.g1"]["hot_block"])) == 0
    assert g2["lds"] >= 48 and g2["scratch_loads"] + g2["scratch_stores"] < 40, g2
    side = os.path.splitext(built_lib)[0] + ".stats.json"
    if os.path.exists(side):
        assert json.load(open(side))["bls12_381.g1"]["hot_block"]["multiply_adds"] == g1["multiply_adds"]
