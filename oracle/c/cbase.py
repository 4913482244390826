"""TEST INFRASTRUCTURE: ctypes wrapper + build recipe for the oracle's C restatement (cbase.c).

Roles: (i) oracle at sizes the Python oracle cannot reach (cross-checked against it in
tests/test_oracle_c.py), (ii) `cpu_baseline` of bench.py ("port": arkworks-algorithm CPU restatement --
Pippenger with arkworks' window rule and per-window threads, radix-2 FFT, libsnark-style witness map --
timed on the host cores of the GPU box).  Never imported by snark_amd.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

from ..fields import CURVES, CurveParams

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD_DIR = os.path.normpath(os.path.join(HERE, "..", "_build"))
LIB = os.path.join(BUILD_DIR, "libcbase.so")
_lib = None


def build(force=False, verbose=False):
    src = [os.path.join(HERE, f) for f in ("cbase.c", "fp_tmpl.h", "fp2_tmpl.h", "grp_tmpl.h")]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(s) for s in src):
        return LIB
    os.makedirs(BUILD_DIR, exist_ok=True)
    cmd = ["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", "-o", LIB, src[0]]
    if verbose:
        print("[oracle.c]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def _u64(v, n):
    return (C.c_uint64 * n)(*[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)])


def lib():
    global _lib
    if _lib is None:
        path = build()
        L = C.CDLL(path)
        for cv in (CURVES["bls12_381"], CURVES["bn254"]):
            nq = cv.fq_limbs64
            Rr, Rq = 1 << 256, 1 << (64 * nq)
            rho = pow(cv.fr_generator, (cv.r - 1) >> cv.two_adicity, cv.r)
            L.cb_init(cv.curve_id, _u64(cv.r, 4), _u64(Rr % cv.r, 4), _u64(Rr * Rr % cv.r, 4),
                      C.c_uint64((-pow(cv.r, -1, 1 << 64)) % (1 << 64)),
                      _u64(cv.q, nq), _u64(Rq % cv.q, nq), _u64(Rq * Rq % cv.q, nq),
                      C.c_uint64((-pow(cv.q, -1, 1 << 64)) % (1 << 64)),
                      cv.r.bit_length(), cv.two_adicity, _u64(rho * Rr % cv.r, 4), _u64(cv.fr_generator * Rr % cv.r, 4))
        L.cb_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(b):
    a = np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray)) else np.ascontiguousarray(b)
    return a.ctypes.data_as(C.c_void_p), a


def ntt(curve: CurveParams, data: bytes, log_n, inverse=False, coset=False) -> bytes:
    a = np.frombuffer(bytearray(data), dtype=np.uint8)
    rc = lib().cb_ntt(curve.curve_id, a.ctypes.data_as(C.c_void_p), log_n, int(inverse), int(coset))
    if rc:
        raise ValueError(rc)
    return a.tobytes()


def msm(curve: CurveParams, group, bases: bytes, scalars: bytes, n) -> bytes:
    psz = curve.fq_bytes * (2 if group == 1 else 4)
    out = np.zeros(psz, dtype=np.uint8)
    pb, k1 = _p(bases if n else b"\0")
    ps, k2 = _p(scalars if n else b"\0")
    lib().cb_msm(curve.curve_id, group, pb, ps, C.c_uint64(n), out.ctypes.data_as(C.c_void_p))
    return out.tobytes()


def fixed_base(curve: CurveParams, group, base: bytes, scalars: bytes, n) -> bytes:
    psz = curve.fq_bytes * (2 if group == 1 else 4)
    out = np.zeros(max(1, n * psz), dtype=np.uint8)
    pb, k1 = _p(base)
    ps, k2 = _p(scalars if n else b"\0")
    lib().cb_fixed_base(curve.curve_id, group, pb, ps, C.c_uint64(n), out.ctypes.data_as(C.c_void_p))
    return out.tobytes()[:n * psz]


def _csr_args(mats):
    args, keep = [], []
    for rp, col, cf in mats:
        a = np.ascontiguousarray(rp, dtype=np.uint64)
        b = np.ascontiguousarray(col, dtype=np.uint32)
        if b.size == 0:
            b = np.zeros(1, dtype=np.uint32)
        c = np.frombuffer(cf if len(cf) else bytes(32), dtype=np.uint8)
        keep += [a, b, c]
        args += [a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)]
    return args, keep


def witness_map(curve: CurveParams, n, ell, w, mats, z: bytes) -> bytes:
    need = n + ell
    N = 1
    while N < need:
        N <<= 1
    out = np.zeros(N * 32, dtype=np.uint8)
    args, keep = _csr_args(mats)
    pz, kz = _p(z)
    rc = lib().cb_witness_map(curve.curve_id, C.c_uint64(n), C.c_uint64(ell), C.c_uint64(w), *args, pz,
                              out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError(rc)
    return out.tobytes()


def points_serialize(curve: CurveParams, group: int, raw, compressed: bool) -> bytes:
    """ark-serialize encoding of a vector of raw affine points (cb_points_serialize), without the length prefix."""
    nb = curve.fq_bytes
    rsz = 2 * nb * (1 if group == 1 else 2)
    raw = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw.view(np.uint8).reshape(-1)
    assert len(raw) % rsz == 0
    n = len(raw) // rsz
    psz = nb * (1 if group == 1 else 2) * (1 if compressed else 2)
    out = np.zeros(max(1, n * psz), dtype=np.uint8)
    src = np.ascontiguousarray(raw) if n else np.zeros(8, dtype=np.uint8)
    rc = lib().cb_points_serialize(curve.curve_id, group, src.ctypes.data_as(C.c_void_p), C.c_uint64(n), int(bool(compressed)),
                                   out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError(rc)
    return out[:n * psz].tobytes()


def pk_stream(curve: CurveParams, pk_raw: dict, compressed: bool) -> bytes:
    """`ProvingKey` byte stream (oracle/serialize.py pk_bytes layout) from the raw key of setup_raw_c."""
    def e(group, raw):
        return points_serialize(curve, group, raw, compressed)

    def vec(group, raw):
        rsz = 2 * curve.fq_bytes * (1 if group == 1 else 2)
        return (len(raw) // rsz).to_bytes(8, "little") + e(group, raw)
    return (e(1, pk_raw["alpha_g1"]) + e(2, pk_raw["beta_g2"]) + e(2, pk_raw["gamma_g2"]) + e(2, pk_raw["delta_g2"])
            + vec(1, pk_raw["gamma_abc_g1"]) + e(1, pk_raw["beta_g1"]) + e(1, pk_raw["delta_g1"])
            + vec(1, pk_raw["a_query"]) + vec(1, pk_raw["b_g1_query"]) + vec(2, pk_raw["b_g2_query"])
            + vec(1, pk_raw["h_query"]) + vec(1, pk_raw["l_query"]))


def prove(curve: CurveParams, n, ell, w, mats, z: bytes, pk_raw: dict, r: int, s: int, timings=None):
    """pk_raw: dict of raw byte arrays with the names of ark355_pk_desc.  Returns (a, b, c) raw affine."""
    g1, g2 = 2 * curve.fq_bytes, 4 * curve.fq_bytes
    oa, ob, oc = (np.zeros(k, dtype=np.uint8) for k in (g1, g2, g1))
    args, keep = _csr_args(mats)
    ptrs = []
    for name in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query", "alpha_g1", "beta_g1", "delta_g1",
                 "beta_g2", "delta_g2"):
        p, k = _p(pk_raw[name] if len(pk_raw[name]) else b"\0")
        keep.append(k)
        ptrs.append(p)
    pz, kz = _p(z)
    tm = (C.c_double * 4)()
    rc = lib().cb_prove(curve.curve_id, C.c_uint64(n), C.c_uint64(ell), C.c_uint64(w), *args, pz, *ptrs,
                        _u64(r % curve.r, 4), _u64(s % curve.r, 4), oa.ctypes.data_as(C.c_void_p),
                        ob.ctypes.data_as(C.c_void_p), oc.ctypes.data_as(C.c_void_p), tm)
    if rc:
        raise ValueError(rc)
    if timings is not None:
        timings.update(witness_map_s=tm[0], msm_s=tm[1], total_s=tm[2])
    return oa.tobytes(), ob.tobytes(), oc.tobytes()


# ---- setup with the C fixed-base routine (valid keys at sizes Python cannot reach) -----------------------------
def setup_raw(curve: CurveParams, A, B, Cm, ell, m, td):
    """Groth16 key as raw byte arrays (names of ark355_pk_desc) + the oracle-side scalars."""
    from .. import groth16 as G, serialize as Z
    from ..curves import g1 as G1of, g2 as G2of
    r = curve.r
    n = len(A)
    u, v, w, zt, dom = G.qap_scalars(curve, A, B, Cm, n, ell, m, td.tau)
    N = dom.n
    gi, di = pow(td.gamma, -1, r), pow(td.delta, -1, r)
    abc = [(td.beta * u[i] + td.alpha * v[i] + w[i]) % r for i in range(m)]
    h_s, t = [], zt * di % r
    for _ in range(N - 1):
        h_s.append(t)
        t = t * td.tau % r
    G1, G2 = G1of(curve), G2of(curve)
    b1 = Z.g1_raw(curve, G1.mul(curve.g1_gen, td.g1_k))
    b2 = Z.g2_raw(curve, G2.mul(curve.g2_gen, td.g2_k))

    def fb(group, ks):
        sb = b"".join((k % r).to_bytes(32, "little") for k in ks)
        return fixed_base(curve, group, b1 if group == 1 else b2, sb, len(ks))

    s1, s2 = 2 * curve.fq_bytes, 4 * curve.fq_bytes
    one1 = fb(1, [td.alpha, td.beta, td.delta])
    one2 = fb(2, [td.beta, td.gamma, td.delta])
    pk = dict(a_query=fb(1, u), b_g1_query=fb(1, v), b_g2_query=fb(2, v), h_query=fb(1, h_s),
              l_query=fb(1, [abc[i] * di % r for i in range(ell, m)]),
              alpha_g1=one1[:s1], beta_g1=one1[s1:2 * s1], delta_g1=one1[2 * s1:],
              beta_g2=one2[:s2], gamma_g2=one2[s2:2 * s2], delta_g2=one2[2 * s2:],
              gamma_abc_g1=fb(1, [abc[i] * gi % r for i in range(ell)]))
    return pk, dict(u=u, v=v, w=w, N=N)


def setup_raw_c(curve: CurveParams, n, ell, w, mats, td):
    """Same key as setup_raw, with the generator's scalars computed by the C restatement (cb_setup_scalars):
    valid keys at 2^18..2^22 constraints in seconds.  `mats`: three (row_ptr u64, col u32, coeff Montgomery bytes).
    Returns (pk_raw dict with the names of ark355_pk_desc plus gamma_g2 / gamma_abc_g1, scalars dict of canonical
    little-endian byte strings u, v, w for the closed form)."""
    from .. import serialize as Z
    from ..curves import g1 as G1of, g2 as G2of
    r = curve.r
    m = ell + w
    N = 1
    while N < n + ell:
        N <<= 1
    args, keep = _csr_args(mats)
    tdb = np.frombuffer(b"".join((x % r).to_bytes(32, "little") for x in (td.tau, td.alpha, td.beta, td.gamma, td.delta)),
                        dtype=np.uint8)
    outs = {k: np.zeros(max(1, cnt) * 32, dtype=np.uint8)
            for k, cnt in (("u", m), ("v", m), ("w", m), ("l", w), ("gabc", ell), ("h", N - 1))}
    rc = lib().cb_setup_scalars(curve.curve_id, C.c_uint64(n), C.c_uint64(ell), C.c_uint64(w), *args,
                                tdb.ctypes.data_as(C.c_void_p),
                                *[outs[k].ctypes.data_as(C.c_void_p) for k in ("u", "v", "w", "l", "gabc", "h")])
    if rc:
        raise ValueError(rc)
    G1, G2 = G1of(curve), G2of(curve)
    b1 = Z.g1_raw(curve, G1.mul(curve.g1_gen, td.g1_k))
    b2 = Z.g2_raw(curve, G2.mul(curve.g2_gen, td.g2_k))

    def fb(group, arr, cnt):
        return fixed_base(curve, group, b1 if group == 1 else b2, arr[:cnt * 32], cnt)

    def fbi(group, ks):
        sb = np.frombuffer(b"".join((k % r).to_bytes(32, "little") for k in ks), dtype=np.uint8)
        return fb(group, sb, len(ks))

    s1, s2 = 2 * curve.fq_bytes, 4 * curve.fq_bytes
    one1 = fbi(1, [td.alpha, td.beta, td.delta])
    one2 = fbi(2, [td.beta, td.gamma, td.delta])
    pk = dict(a_query=fb(1, outs["u"], m), b_g1_query=fb(1, outs["v"], m), b_g2_query=fb(2, outs["v"], m),
              h_query=fb(1, outs["h"], N - 1), l_query=fb(1, outs["l"], w),
              alpha_g1=one1[:s1], beta_g1=one1[s1:2 * s1], delta_g1=one1[2 * s1:],
              beta_g2=one2[:s2], gamma_g2=one2[s2:2 * s2], delta_g2=one2[2 * s2:],
              gamma_abc_g1=fb(1, outs["gabc"], ell))
    return pk, dict(u=outs["u"][:m * 32].tobytes(), v=outs["v"][:m * 32].tobytes(), w=outs["w"][:m * 32].tobytes(), N=N)


def _cpu_quota():
    """CPU bandwidth limit of the container in cores (cgroup v2 cpu.max or v1 cfs quota); None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return float(q) / float(per)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except Exception:
        pass
    return None


def _bench_prove_workload(cv, curve_name, L, hw, quota, cores, log_n, budget_s, wl):
    """bench_prove on the caller's own statement and key; the first proof doubles as the parity check of the caller's proof."""
    n, ell, w, mats, zb, pk = wl["n"], wl["ell"], wl["w"], wl["mats"], wl["z"], wl["pk"]
    N = 1
    while N < n + ell:
        N <<= 1
    rng = np.random.default_rng(1)
    times, t_start = [], time.perf_counter()
    tm = {}
    got = prove(cv, n, ell, w, mats, zb, pk, wl["r"], wl["s"], timings=tm)
    parity = "GPU proof bytes == oracle/c proof bytes (same statement, key, z, r, s)" if tuple(got) == tuple(wl["proof"]) else "MISMATCH"
    times.append((tm["total_s"], tm["witness_map_s"], tm["msm_s"]))
    while len(times) < 5 and (len(times) < 2 or (time.perf_counter() - t_start) < budget_s):
        tm = {}
        prove(cv, n, ell, w, mats, zb, pk, int(rng.integers(1, 1 << 62)), int(rng.integers(1, 1 << 62)), timings=tm)
        times.append((tm["total_s"], tm["witness_map_s"], tm["msm_s"]))
    L.cb_set_threads(hw)
    times.sort()
    med, med_w, med_m = times[len(times) // 2]
    return {"value": n / med, "unit": "constraints/s", "cores": cores, "kind": "port", "parity_vs_oracle": parity,
            "sample": "oracle/c (arkworks-algorithm C restatement; Pippenger tasks = window x term-chunk, OpenMP x%d): "
                      "median of %d Groth16/%s proofs of THE BENCH'S OWN statement and key (S2 mulchain, n = %d, N = 2^%d), "
                      "%.3f s each (witness map %.3f s, MSMs + tail %.3f s), assignment in host memory -> proof%s"
                      % (cores, len(times), curve_name, n, N.bit_length() - 1, med, med_w, med_m,
                         ("; host shows %d hardware threads, container CPU quota %.1f cores" % (hw, quota)) if quota else "")}


def bench_prove(curve_name="bls12_381", log_n=None, budget_s=20.0, workload=None):
    """cpu_baseline for bench.py: the C restatement on all host cores, on the benchmark's own configuration (S2
    mulchain, n = 2^20, N = 2^21) whenever the host has the cores to finish a proof in seconds (>= 12 threads);
    smaller hosts time a 2^16 sample and say so.  Same timing window as the GPU figure: assignment resident in host
    memory -> three affine proof points.  Bases are k_i*G made with the C fixed-base routine
    (their distribution does not affect Pippenger's cost).

    workload (round 6; the oracle as CHECKER of the bench's own proofs): dict(n, ell, w, mats, z = Montgomery bytes, pk = raw
    key arrays under the names of ark355_pk_desc, r, s, proof = (a, b, c) raw affine bytes the GPU returned for (z, r, s)).
    When the host can time the full size, the timed CPU proofs run on THIS statement and key, the first of them with the
    bench's (r, s), and its bytes are compared with the GPU's: the line then reports `parity_vs_oracle`."""
    from .. import synthetic as S
    from .. import serialize as Z
    cv = CURVES[curve_name]
    L = lib()
    hw = L.cb_num_threads()
    quota = _cpu_quota()
    # more runnable threads than the container's CPU bandwidth only buys throttling: the MI355X boxes of this pool show
    # 256 hardware threads and cpu.max = 16 cores
    cores = max(1, min(hw, int(quota + 0.5))) if quota else hw
    L.cb_set_threads(cores)
    if log_n is None:
        log_n = 20 if cores >= 12 else 16
    if workload is not None and workload["n"] == (1 << log_n):
        return _bench_prove_workload(cv, curve_name, L, hw, quota, cores, log_n, budget_s, workload)
    n, ell, w, mats, z = S.mulchain_csr(cv.r, 1 << log_n)
    m = ell + w
    N = 1
    while N < n + ell:
        N <<= 1
    rng = np.random.default_rng(1)

    def rand_pts(group, k):
        sc = np.zeros((k, 4), dtype="<u8")
        sc[:, 0] = rng.integers(1, 1 << 63, size=k, dtype=np.uint64)
        base = Z.g1_raw(cv, cv.g1_gen) if group == 1 else Z.g2_raw(cv, cv.g2_gen)
        return fixed_base(cv, group, base, sc.tobytes(), k)

    pk = dict(a_query=rand_pts(1, m), b_g1_query=rand_pts(1, m), b_g2_query=rand_pts(2, m),
              h_query=rand_pts(1, N - 1), l_query=rand_pts(1, w))
    one1, one2 = rand_pts(1, 3), rand_pts(2, 2)
    s1, s2 = 2 * cv.fq_bytes, 4 * cv.fq_bytes
    pk.update(alpha_g1=one1[:s1], beta_g1=one1[s1:2 * s1], delta_g1=one1[2 * s1:], beta_g2=one2[:s2], delta_g2=one2[s2:])
    zb = S._mont_bytes(cv.r, z)
    prove(cv, n, ell, w, mats, zb, pk, 3, 5)                      # warm-up
    times, t_start = [], time.perf_counter()
    while len(times) < 5 and (len(times) < 2 or (time.perf_counter() - t_start) < budget_s):
        tm = {}
        prove(cv, n, ell, w, mats, zb, pk, int(rng.integers(1, 1 << 62)), int(rng.integers(1, 1 << 62)), timings=tm)
        times.append((tm["total_s"], tm["witness_map_s"], tm["msm_s"]))
    L.cb_set_threads(hw)
    times.sort()
    med, med_w, med_m = times[len(times) // 2]
    return {"value": n / med, "unit": "constraints/s", "cores": cores, "kind": "port",
            "sample": "oracle/c (arkworks-algorithm C restatement; Pippenger tasks = window x term-chunk, OpenMP x%d): "
                      "median of %d Groth16/%s proofs of the S2 mulchain R1CS with n = 2^%d constraints (N = 2^%d), "
                      "%.3f s each (witness map %.3f s, MSMs + tail %.3f s), assignment in host memory -> proof%s"
                      % (cores, len(times), curve_name, log_n, N.bit_length() - 1, med, med_w, med_m,
                         ("; host shows %d hardware threads, container CPU quota %.1f cores" % (hw, quota)) if quota else "")}
