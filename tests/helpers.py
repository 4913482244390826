"""Shared test plumbing: oracle objects <-> the raw memory images of the C ABI."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import serialize as Z  # noqa: E402
from oracle.fields import CurveParams  # noqa: E402


def csr_from_rows(curve: CurveParams, rows):
    """Matrix<F> = Vec<Vec<(F, usize)>> -> (row_ptr u64, col u32, coeff bytes Montgomery)."""
    row_ptr = np.zeros(len(rows) + 1, dtype=np.uint64)
    cols, coeffs = [], []
    k = 0
    for i, row in enumerate(rows):
        for c, j in row:
            cols.append(j)
            coeffs.append(Z.fr_mont(curve, c))
            k += 1
        row_ptr[i + 1] = k
    return row_ptr, np.array(cols, dtype=np.uint32), b"".join(coeffs)


def z_bytes(curve, z):
    return b"".join(Z.fr_mont(curve, v) for v in z)


def fr_vec_from_mont(curve, b):
    nb = curve.fr_bytes
    return [Z.fr_from_mont(curve, b[i * nb:(i + 1) * nb]) for i in range(len(b) // nb)]


def g1_vec_raw(curve, pts):
    return b"".join(Z.g1_raw(curve, p) for p in pts)


def g2_vec_raw(curve, pts):
    return b"".join(Z.g2_raw(curve, p) for p in pts)


def pk_load_from_oracle(lib, ctx, curve, pk, ell, w, N):
    return lib.pk_load(
        ctx, curve.curve_id, ell, w, N,
        g1_vec_raw(curve, pk.a_query), g1_vec_raw(curve, pk.b_g1_query), g2_vec_raw(curve, pk.b_g2_query),
        g1_vec_raw(curve, pk.h_query), g1_vec_raw(curve, pk.l_query),
        Z.g1_raw(curve, pk.vk.alpha_g1), Z.g1_raw(curve, pk.beta_g1), Z.g1_raw(curve, pk.delta_g1),
        Z.g2_raw(curve, pk.vk.beta_g2), Z.g2_raw(curve, pk.vk.delta_g2))


def r1cs_load_from_rows(lib, ctx, curve, A, B, Cm, ell, w):
    mats = [csr_from_rows(curve, M) for M in (A, B, Cm)]
    return lib.r1cs_load(ctx, curve.curve_id, len(A), ell, w, mats)
