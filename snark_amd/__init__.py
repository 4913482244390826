"""snark_amd -- MI355X (gfx950) Groth16 prover backend for the arkworks SNARK / R1CS trait surface.

The product is ``libark355.so`` (hand-written HIP for CDNA4 behind the C ABI of ``include/ark355.h``).
This package is the thin host-side image of that ABI plus the Python mirror of the reference's
``SNARK`` trait for this one path (``snark_amd.groth16``).  There is **no CPU fallback**: importing
``lib()`` without the built extension raises.
"""
from __future__ import annotations

import os

from . import _binding
from ._binding import (BLS12_381, BN254, Ark355Error, Lib, OK, EINVAL, ENOMEM, EHIP, ENODEV,  # noqa: F401
                       E_ASSIGNMENT_MISSING, E_UNSATISFIABLE, E_POLY_DEGREE_TOO_LARGE)

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARK355_LIB: developer override to A/B-test another BUILD OF THE SAME HIP LIBRARY (still no fallback of any kind)
LIB_PATH = os.environ.get("ARK355_LIB") or os.path.join(_HERE, "libark355.so")
_lib = None


def lib() -> Lib:
    """The loaded HIP library.  Fails loudly when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "snark_amd: %s is missing -- build it with `python -m snark_amd.build` "
                "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
        _lib = Lib(LIB_PATH)
    return _lib


CURVE_IDS = {"bls12_381": BLS12_381, "bn254": BN254}
