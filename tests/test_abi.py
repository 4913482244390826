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
