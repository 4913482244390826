"""ctypes binding of the C ABI declared in ``include/ark355.h``.

This is the Python-side image of the ``extern "C"`` block a Rust maintainer would write (see
INTEGRATION.md).  It is deliberately free of torch: plain pointers and sizes only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

BLS12_381 = 0
BN254 = 1

OK = 0
EINVAL = -1
ENOMEM = -2
EHIP = -3
ERCCL = -4
ENODEV = -5
E_ASSIGNMENT_MISSING = -16
E_UNSATISFIABLE = -17
E_POLY_DEGREE_TOO_LARGE = -18
COMM_ID_BYTES = 128
SHARD_WINDOW = 0
SHARD_BUCKET_RING = 1

_ERR_NAMES = {
    EINVAL: "EINVAL", ENOMEM: "ENOMEM", EHIP: "EHIP", ERCCL: "ERCCL", ENODEV: "ENODEV",
    E_ASSIGNMENT_MISSING: "AssignmentMissing", E_UNSATISFIABLE: "Unsatisfiable",
    E_POLY_DEGREE_TOO_LARGE: "PolynomialDegreeTooLarge",
}

# every symbol include/ark355.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "ark355_ctx_create", "ark355_ctx_destroy", "ark355_last_error", "ark355_version", "ark355_sizes",
    "ark355_host_alloc", "ark355_host_free",
    "ark355_pk_load", "ark355_pk_free", "ark355_r1cs_load", "ark355_r1cs_free",
    "ark355_r1cs_domain_size", "ark355_prove", "ark355_prove_dev", "ark355_witness_map", "ark355_witness_map_dist_sim",
    "ark355_is_satisfied", "ark355_r1cs_mat_vec", "ark355_ntt_fr", "ark355_ntt_fr_dev",
    "ark355_msm_g1", "ark355_msm_g2", "ark355_bases_load", "ark355_bases_free", "ark355_msm_dev",
    "ark355_msm_dev_partial", "ark355_xyzz_sum", "ark355_fixed_base_mul", "ark355_get_timings",
    "ark355_get_kernel_stats", "ark355_pk_load_shard", "ark355_partial_size", "ark355_prove_shard",
    "ark355_prove_combine", "ark355_prove_batch", "ark355_comm_unique_id", "ark355_comm_init", "ark355_comm_destroy",
    "ark355_prove_sharded", "ark355_prove_sharded_dev", "ark355_point_size", "ark355_pk_load_bytes", "ark355_pk_dims",
    "ark355_pk_table_info",
    "ark355_points_decode", "ark355_points_encode", "ark355_proof_to_bytes", "ark355_proof_from_bytes",
    "ark355_setup_scalars", "ark355_verify_batch",
    "ark355_ctx_set_policy", "ark355_ctx_get_policy", "ark355_sched_info", "ark355_sched_reset", "ark355_diag_streams", "ark355_diag_dispatch",
    "ark355_diag_mad_rate", "ark355_diag_clocks",
]

SCHED_NAMES = {-1: "auto", 0: "one_stream", 1: "pipeline", 2: "pipeline_sync", 3: "one_stream_spin"}


class Ark355Error(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("%s (%d)%s" % (_ERR_NAMES.get(code, "error"), code, (": " + msg) if msg else ""))


class PkDesc(C.Structure):
    _fields_ = [
        ("num_instance", C.c_uint64), ("num_witness", C.c_uint64), ("domain_size", C.c_uint64),
        ("a_query", C.c_void_p), ("b_g1_query", C.c_void_p), ("b_g2_query", C.c_void_p),
        ("h_query", C.c_void_p), ("l_query", C.c_void_p),
        ("alpha_g1", C.c_void_p), ("beta_g1", C.c_void_p), ("delta_g1", C.c_void_p),
        ("beta_g2", C.c_void_p), ("delta_g2", C.c_void_p),
    ]


class VkDesc(C.Structure):
    _fields_ = [("num_instance", C.c_uint64), ("alpha_g1", C.c_void_p), ("beta_g2", C.c_void_p), ("gamma_g2", C.c_void_p),
                ("delta_g2", C.c_void_p), ("gamma_abc_g1", C.c_void_p)]


class ProofRaw(C.Structure):
    _fields_ = [("a", C.c_uint8 * 96), ("b", C.c_uint8 * 192), ("c", C.c_uint8 * 96)]


class SchedReport(C.Structure):
    _fields_ = [("latched", C.c_int32), ("last", C.c_int32), ("samples", C.c_uint32 * 4), ("mean_ms", C.c_double * 4)]


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "total_ms", "h2d_ms", "witness_map_ms", "msm_h_ms", "msm_l_ms", "msm_ab_g1_ms", "msm_b_g2_ms",
        "finalize_ms")]


def _buf(b):
    """bytes / bytearray / numpy array -> (ctypes pointer, keepalive)."""
    if b is None:
        return None, None
    if isinstance(b, np.ndarray):
        a = np.ascontiguousarray(b)
        return a.ctypes.data_as(C.c_void_p), a
    if isinstance(b, (bytes, bytearray, memoryview)):
        a = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, bytearray) else np.frombuffer(b, dtype=np.uint8)
        return a.ctypes.data_as(C.c_void_p), a
    raise TypeError(type(b))


class Lib:
    """Loaded libark355 with typed signatures and thin call helpers."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.dll = C.CDLL(path)          # RTLD_LOCAL: the emulator test build exports the same symbol names
        d = self.dll
        vp, u64, u32, i32, i64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_int64
        P = C.POINTER
        d.ark355_ctx_create.argtypes = [i32, P(vp)]
        d.ark355_ctx_destroy.argtypes = [vp]
        d.ark355_ctx_destroy.restype = None
        d.ark355_last_error.argtypes = [vp]
        d.ark355_last_error.restype = C.c_char_p
        d.ark355_version.restype = u32
        d.ark355_sizes.argtypes = [i32, P(u32 * 4)]
        d.ark355_host_alloc.argtypes = [u64, P(vp)]
        d.ark355_host_free.argtypes = [vp]
        d.ark355_host_free.restype = None
        d.ark355_pk_load.argtypes = [vp, i32, P(PkDesc), P(vp)]
        d.ark355_pk_free.argtypes = [vp]
        d.ark355_pk_free.restype = None
        d.ark355_r1cs_load.argtypes = [vp, i32, u64, u64, u64, P(vp * 3), P(vp * 3), P(vp * 3), P(vp)]
        d.ark355_r1cs_free.argtypes = [vp]
        d.ark355_r1cs_free.restype = None
        d.ark355_r1cs_domain_size.argtypes = [vp]
        d.ark355_r1cs_domain_size.restype = u64
        d.ark355_prove.argtypes = [vp, vp, vp, vp, u64, vp, vp, P(ProofRaw)]
        d.ark355_prove_dev.argtypes = [vp, vp, vp, vp, u64, vp, vp, P(ProofRaw)]
        d.ark355_witness_map.argtypes = [vp, vp, vp, u64, vp]
        d.ark355_witness_map_dist_sim.argtypes = [vp, vp, vp, u64, C.c_uint32, vp]
        d.ark355_is_satisfied.argtypes = [vp, vp, vp, u64, P(i64)]
        d.ark355_r1cs_mat_vec.argtypes = [vp, vp, vp, u64, vp, vp, vp]
        d.ark355_ntt_fr.argtypes = [vp, i32, vp, u32, i32, i32]
        d.ark355_ntt_fr_dev.argtypes = [vp, i32, vp, vp, u32, i32, i32, vp]
        d.ark355_msm_g1.argtypes = [vp, i32, vp, vp, u64, vp]
        d.ark355_msm_g2.argtypes = [vp, i32, vp, vp, u64, vp]
        d.ark355_bases_load.argtypes = [vp, i32, i32, vp, u64, P(vp)]
        d.ark355_bases_free.argtypes = [vp]
        d.ark355_bases_free.restype = None
        d.ark355_msm_dev.argtypes = [vp, vp, vp, u64, i32, vp]
        d.ark355_msm_dev_partial.argtypes = [vp, vp, vp, u64, i32, vp]
        d.ark355_xyzz_sum.argtypes = [vp, i32, i32, vp, u64, vp]
        d.ark355_fixed_base_mul.argtypes = [vp, i32, i32, vp, vp, u64, vp]
        d.ark355_pk_load_shard.argtypes = [vp, i32, P(PkDesc), u32, u32, P(vp)]
        d.ark355_partial_size.argtypes = [i32]
        d.ark355_partial_size.restype = u64
        d.ark355_prove_shard.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp]
        d.ark355_prove_combine.argtypes = [vp, i32, vp, u64, vp, vp, P(ProofRaw)]
        d.ark355_prove_batch.argtypes = [vp, vp, vp, vp, u64, vp, vp, u64, u32, vp]
        d.ark355_comm_unique_id.argtypes = [vp]
        d.ark355_comm_init.argtypes = [vp, vp, i32, i32, P(vp)]
        d.ark355_comm_destroy.argtypes = [vp]
        d.ark355_comm_destroy.restype = None
        d.ark355_prove_sharded.argtypes = [vp, vp, vp, vp, vp, u64, vp, vp, i32, P(ProofRaw)]
        d.ark355_prove_sharded_dev.argtypes = [vp, vp, vp, vp, vp, u64, vp, vp, i32, P(ProofRaw)]
        d.ark355_point_size.argtypes = [i32, i32, i32]
        d.ark355_point_size.restype = u64
        d.ark355_pk_load_bytes.argtypes = [vp, i32, vp, u64, i32, i32, P(vp)]
        d.ark355_pk_dims.argtypes = [vp, P(u64), P(u64), P(u64)]
        d.ark355_pk_table_info.argtypes = [vp, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32), P(u64)]
        d.ark355_points_decode.argtypes = [vp, i32, i32, vp, u64, i32, i32, vp]
        d.ark355_points_encode.argtypes = [vp, i32, i32, vp, u64, i32, vp]
        d.ark355_proof_to_bytes.argtypes = [i32, P(ProofRaw), i32, vp]
        d.ark355_proof_from_bytes.argtypes = [i32, vp, u64, i32, i32, P(ProofRaw)]
        d.ark355_setup_scalars.argtypes = [i32, u64, u64, u64, P(vp * 3), P(vp * 3), P(vp * 3), vp, vp, vp, vp, vp, vp, vp]
        d.ark355_verify_batch.argtypes = [vp, i32, P(VkDesc), vp, vp, vp, u64, P(i32)]
        d.ark355_ctx_set_policy.argtypes = [vp, C.c_char_p, i64]
        d.ark355_ctx_get_policy.argtypes = [vp, C.c_char_p, P(i64)]
        d.ark355_sched_info.argtypes = [vp, vp, i32, P(SchedReport)]
        d.ark355_sched_reset.argtypes = [vp]
        d.ark355_diag_streams.argtypes = [vp, u32, vp]
        d.ark355_diag_dispatch.argtypes = [vp, u32, u32, P(C.c_float), P(u32)]
        d.ark355_diag_mad_rate.argtypes = [vp, C.c_float, P(C.c_float), P(C.c_float)]
        d.ark355_diag_clocks.argtypes = [vp, P(u64), u32, P(u32)]
        d.ark355_get_timings.argtypes = [vp, P(Timings)]
        d.ark355_get_kernel_stats.argtypes = [vp, P(C.c_float), P(u64), P(u64)]
        for name in SYMBOLS:
            fn = getattr(d, name)
            if fn.restype is C.c_int:      # default: make every status-returning call int32
                fn.restype = i32

    # ---- runtime policy (include/ark355.h "runtime policy") ------------------------------------------
    def ctx_set_policy(self, ctx, name, value):
        self.check(ctx, self.dll.ark355_ctx_set_policy(ctx, name.encode(), int(value)))

    def ctx_get_policy(self, ctx, name):
        v = C.c_int64()
        self.check(ctx, self.dll.ark355_ctx_get_policy(ctx, name.encode(), C.byref(v)))
        return int(v.value)

    def policy(self, ctx, **kw):
        """context manager: set policy values on `ctx`, restore the previous ones on exit"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = {k: self.ctx_get_policy(ctx, k) for k in kw}
            try:
                for k, v in kw.items():
                    self.ctx_set_policy(ctx, k, v)
                yield
            finally:
                for k, v in old.items():
                    self.ctx_set_policy(ctx, k, v)
        return cm()

    def sched_info(self, ctx, pk, in_flight):
        r = SchedReport()
        self.check(ctx, self.dll.ark355_sched_info(ctx, pk, 1 if in_flight else 0, C.byref(r)))
        return {"latched": SCHED_NAMES.get(r.latched, str(r.latched)), "last": SCHED_NAMES.get(r.last, str(r.last)),
                "samples": {SCHED_NAMES[i]: int(r.samples[i]) for i in range(4) if r.samples[i]},
                "mean_ms": {SCHED_NAMES[i]: round(float(r.mean_ms[i]), 3) for i in range(4) if r.samples[i]}}

    def diag_streams(self, ctxs):
        """ark355_diag_streams: rows of the serialisation matrix over [ctx streams..., sW, sS, sR of ctxs[0]]"""
        n = len(ctxs) + 3
        arr = (C.c_void_p * len(ctxs))(*ctxs)
        out = (C.c_int8 * (n * n))()
        self.check(ctxs[0], self.dll.ark355_diag_streams(arr, len(ctxs), out))
        return [[int(out[i * n + j]) for j in range(n)] for i in range(n)]

    def diag_dispatch(self, ctx, launches=200, spin_us=20):
        gap, lanes = C.c_float(0), C.c_uint32(0)
        self.check(ctx, self.dll.ark355_diag_dispatch(ctx, launches, spin_us, C.byref(gap), C.byref(lanes)))
        return {"gap_us": round(float(gap.value), 2), "launches": launches, "spin_us": spin_us, "lanes": int(lanes.value)}

    def diag_mad_rate(self, ctx, target_ms=20.0):
        """T (10^12) v_mad_u64_u32 per second this box issues right now at the accumulation kernels' occupancy (< 0: emulator)."""
        rate, ms = C.c_float(0), C.c_float(0)
        self.check(ctx, self.dll.ark355_diag_mad_rate(ctx, float(target_ms), C.byref(rate), C.byref(ms)))
        return {"tmad_per_s": float(rate.value), "elapsed_ms": float(ms.value)}

    def diag_clocks(self, ctx):
        """{slot: (shader-clock cycles, ticks of the constant 100 MHz reference)} read on the GPU, one pair per compute unit."""
        cap = 1024
        buf = (C.c_uint64 * (2 * cap))()
        cnt = C.c_uint32(0)
        self.check(ctx, self.dll.ark355_diag_clocks(ctx, buf, cap, C.byref(cnt)))
        return {i: (int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(int(cnt.value)) if buf[2 * i + 1]}

    @staticmethod
    def diag_clocks_delta(c0, c1):
        """Two diag_clocks readings -> (median gfx clock in MHz over the compute units present in both, elapsed ms, units used)."""
        ratios, ticks = [], []
        for k, (a0, r0) in c0.items():
            if k in c1:
                a1, r1 = c1[k]
                if a1 > a0 and r1 > r0:
                    ratios.append((a1 - a0) / (r1 - r0) * 100.0)
                    ticks.append(r1 - r0)
        if not ratios:
            return None, None, 0
        ratios.sort()
        ticks.sort()
        return ratios[len(ratios) // 2], ticks[len(ticks) // 2] / 1e5, len(ratios)

    def sched_reset(self, ctx):
        self.check(ctx, self.dll.ark355_sched_reset(ctx))

    # ---- helpers ---------------------------------------------------------------------------------
    def check(self, ctx, rc):
        if rc != OK:
            msg = ""
            if ctx:
                m = self.dll.ark355_last_error(ctx)
                msg = m.decode() if m else ""
            raise Ark355Error(rc, msg)

    def sizes(self, curve):
        out = (C.c_uint32 * 4)()
        self.check(None, self.dll.ark355_sizes(curve, C.byref(out)))
        return {"fr": out[0], "fq": out[1], "g1": out[2], "g2": out[3]}

    def ctx_create(self, device=0):
        h = C.c_void_p()
        rc = self.dll.ark355_ctx_create(device, C.byref(h))
        if rc != OK:
            raise Ark355Error(rc, "ark355_ctx_create")
        return h

    def ctx_destroy(self, ctx):
        self.dll.ark355_ctx_destroy(ctx)

    def pk_load(self, ctx, curve, ell, w, N, a_query, b_g1_query, b_g2_query, h_query, l_query,
                alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2, shard=None):
        keep = []
        d = PkDesc()
        d.num_instance, d.num_witness, d.domain_size = ell, w, N
        for name, val in (("a_query", a_query), ("b_g1_query", b_g1_query), ("b_g2_query", b_g2_query),
                          ("h_query", h_query), ("l_query", l_query), ("alpha_g1", alpha_g1),
                          ("beta_g1", beta_g1), ("delta_g1", delta_g1), ("beta_g2", beta_g2),
                          ("delta_g2", delta_g2)):
            p, k = _buf(val if (val is None or len(val)) else None)
            keep.append(k)
            setattr(d, name, p.value if p is not None else None)
        h = C.c_void_p()
        if shard is None:
            self.check(ctx, self.dll.ark355_pk_load(ctx, curve, C.byref(d), C.byref(h)))
        else:
            self.check(ctx, self.dll.ark355_pk_load_shard(ctx, curve, C.byref(d), shard[0], shard[1], C.byref(h)))
        return h

    def r1cs_load(self, ctx, curve, n, ell, w, mats):
        """mats: three (row_ptr u64[n+1], col u32[nnz], coeff bytes nnz*fr) tuples."""
        keep = []
        rp = (C.c_void_p * 3)()
        cl = (C.c_void_p * 3)()
        cf = (C.c_void_p * 3)()
        for i, (row_ptr, col, coeff) in enumerate(mats):
            a = np.ascontiguousarray(row_ptr, dtype=np.uint64)
            b = np.ascontiguousarray(col, dtype=np.uint32)
            c, kc = _buf(coeff if len(coeff) else b"\0")
            if b.size == 0:
                b = np.zeros(1, np.uint32)          # a valid pointer for an empty matrix (never read: nnz == 0)
            keep += [a, b, kc]
            rp[i] = a.ctypes.data
            cl[i] = b.ctypes.data
            cf[i] = c.value
        h = C.c_void_p()
        self.check(ctx, self.dll.ark355_r1cs_load(ctx, curve, n, ell, w, C.byref(rp), C.byref(cl), C.byref(cf),
                                                  C.byref(h)))
        return h

    def prove(self, ctx, pk, r1cs, z, z_len, r: bytes, s: bytes, sizes, z_is_device_ptr=False):
        out = ProofRaw()
        rb, k1 = _buf(r)
        sb, k2 = _buf(s)
        if z_is_device_ptr:
            rc = self.dll.ark355_prove_dev(ctx, pk, r1cs, C.c_void_p(z), z_len, rb, sb, C.byref(out))
        else:
            zb, k3 = _buf(z)
            rc = self.dll.ark355_prove(ctx, pk, r1cs, zb, z_len, rb, sb, C.byref(out))
        self.check(ctx, rc)
        return bytes(out.a)[:sizes["g1"]], bytes(out.b)[:sizes["g2"]], bytes(out.c)[:sizes["g1"]]

    def prove_batch(self, ctx, pk, r1cs, zs, z_len, rs, ss, sizes, inflight=3):
        """ark355_prove_batch: `zs` host assignments (bytes), `rs`/`ss` canonical 32-byte scalars, one per proof."""
        count = len(zs)
        out = (ProofRaw * max(1, count))()
        bufs = [_buf(z) for z in zs]
        ptrs = (C.c_void_p * max(1, count))(*[C.cast(b[0], C.c_void_p) for b in bufs])
        rb, k1 = _buf(b"".join(rs) if count else None)
        sb, k2 = _buf(b"".join(ss) if count else None)
        rc = self.dll.ark355_prove_batch(ctx, pk, r1cs, ptrs, z_len, rb, sb, count, int(inflight), out)
        self.check(ctx, rc)
        return [(bytes(o.a)[:sizes["g1"]], bytes(o.b)[:sizes["g2"]], bytes(o.c)[:sizes["g1"]]) for o in out[:count]]

    def prove_shard(self, ctx, curve, pk_shard, r1cs, z, z_len, r: bytes, s: bytes):
        out = np.zeros(self.dll.ark355_partial_size(curve), dtype=np.uint8)
        zb, k1 = _buf(z)
        rb, k2 = _buf(r)
        sb, k3 = _buf(s)
        self.check(ctx, self.dll.ark355_prove_shard(ctx, pk_shard, r1cs, zb, z_len, rb, sb, out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()

    def prove_combine(self, ctx, curve, partials: bytes, count, r: bytes, s: bytes, sizes):
        out = ProofRaw()
        pb, k1 = _buf(partials)
        rb, k2 = _buf(r)
        sb, k3 = _buf(s)
        self.check(ctx, self.dll.ark355_prove_combine(ctx, curve, pb, count, rb, sb, C.byref(out)))
        return bytes(out.a)[:sizes["g1"]], bytes(out.b)[:sizes["g2"]], bytes(out.c)[:sizes["g1"]]

    # ---- ark-serialize wire formats behind the ABI ------------------------------------------------------------
    def point_size(self, curve, group, compressed):
        return int(self.dll.ark355_point_size(curve, group, int(bool(compressed))))

    def pk_load_bytes(self, ctx, curve, data: bytes, compressed=False, validate=True):
        h = C.c_void_p()
        db, k = _buf(data)
        self.check(ctx, self.dll.ark355_pk_load_bytes(ctx, curve, db, len(data), int(bool(compressed)), int(validate),
                                                      C.byref(h)))
        return h

    def pk_table_info(self, pk):
        c, w, st, by = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
        self.check(None, self.dll.ark355_pk_table_info(pk, C.byref(c), C.byref(w), C.byref(st), C.byref(by)))
        return {"window_bits": c.value, "windows": w.value, "table_stride": st.value, "table_bytes": by.value}

    def pk_dims(self, pk):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self.check(None, self.dll.ark355_pk_dims(pk, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def points_decode(self, ctx, curve, group, data: bytes, n, compressed, validate, raw_size):
        out = np.zeros(max(1, n * raw_size), dtype=np.uint8)
        db, k = _buf(data if n else None)
        self.check(ctx, self.dll.ark355_points_decode(ctx, curve, group, db, n, int(bool(compressed)), int(validate),
                                                      out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()[:n * raw_size]

    def points_encode(self, ctx, curve, group, raw: bytes, n, compressed):
        sz = self.point_size(curve, group, compressed)
        out = np.zeros(max(1, n * sz), dtype=np.uint8)
        rb, k = _buf(raw if n else None)
        self.check(ctx, self.dll.ark355_points_encode(ctx, curve, group, rb, n, int(bool(compressed)),
                                                      out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()[:n * sz]

    def proof_to_bytes(self, curve, a: bytes, b: bytes, c: bytes, compressed=True) -> bytes:
        p = ProofRaw()
        C.memmove(p.a, a, len(a))
        C.memmove(p.b, b, len(b))
        C.memmove(p.c, c, len(c))
        n = 2 * self.point_size(curve, 1, compressed) + self.point_size(curve, 2, compressed)
        out = np.zeros(n, dtype=np.uint8)
        self.check(None, self.dll.ark355_proof_to_bytes(curve, C.byref(p), int(bool(compressed)), out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()

    def proof_from_bytes(self, curve, data: bytes, sizes, compressed=True, validate=True):
        p = ProofRaw()
        db, k = _buf(data)
        rc = self.dll.ark355_proof_from_bytes(curve, db, len(data), int(bool(compressed)), int(validate), C.byref(p))
        if rc != OK:
            raise Ark355Error(rc, "ark355_proof_from_bytes")
        return bytes(p.a)[:sizes["g1"]], bytes(p.b)[:sizes["g2"]], bytes(p.c)[:sizes["g1"]]

    # ---- RCCL behind the ABI ----------------------------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        out = np.zeros(COMM_ID_BYTES, dtype=np.uint8)
        self.check(None, self.dll.ark355_comm_unique_id(out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()

    def comm_init(self, ctx, comm_id: bytes, rank: int, world: int):
        h = C.c_void_p()
        ib, k = _buf(comm_id)
        self.check(ctx, self.dll.ark355_comm_init(ctx, ib, rank, world, C.byref(h)))
        return h

    def comm_destroy(self, comm):
        self.dll.ark355_comm_destroy(comm)

    def prove_sharded(self, ctx, comm, pk_shard, r1cs, z, z_len, r: bytes, s: bytes, sizes, mode=SHARD_WINDOW,
                      z_is_device_ptr=False):
        out = ProofRaw()
        rb, k2 = _buf(r)
        sb, k3 = _buf(s)
        if z_is_device_ptr:
            rc = self.dll.ark355_prove_sharded_dev(ctx, comm, pk_shard, r1cs, C.c_void_p(z), z_len, rb, sb, int(mode),
                                                   C.byref(out))
        else:
            zb, k1 = _buf(z)
            rc = self.dll.ark355_prove_sharded(ctx, comm, pk_shard, r1cs, zb, z_len, rb, sb, int(mode), C.byref(out))
        self.check(ctx, rc)
        return bytes(out.a)[:sizes["g1"]], bytes(out.b)[:sizes["g2"]], bytes(out.c)[:sizes["g1"]]

    def witness_map(self, ctx, r1cs, z, z_len, fr_size):
        N = self.dll.ark355_r1cs_domain_size(r1cs)
        out = np.zeros(N * fr_size, dtype=np.uint8)
        zb, k = _buf(z)
        self.check(ctx, self.dll.ark355_witness_map(ctx, r1cs, zb, z_len, out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()

    def witness_map_dist_sim(self, ctx, r1cs, z, z_len, fr_size, world):
        """h as the `world` ranks of a sharded proof compute it, all ranks on this device (test / diagnostic entry)"""
        N = self.dll.ark355_r1cs_domain_size(r1cs)
        out = np.zeros(N * fr_size, dtype=np.uint8)
        zb, k = _buf(z)
        self.check(ctx, self.dll.ark355_witness_map_dist_sim(ctx, r1cs, zb, z_len, int(world), out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()

    def is_satisfied(self, ctx, r1cs, z, z_len):
        fb = C.c_int64(0)
        zb, k = _buf(z)
        self.check(ctx, self.dll.ark355_is_satisfied(ctx, r1cs, zb, z_len, C.byref(fb)))
        return fb.value

    def mat_vec(self, ctx, r1cs, z, z_len, n, fr_size):
        outs = [np.zeros(max(1, n * fr_size), dtype=np.uint8) for _ in range(3)]
        zb, k = _buf(z)
        self.check(ctx, self.dll.ark355_r1cs_mat_vec(ctx, r1cs, zb, z_len, *[o.ctypes.data_as(C.c_void_p) for o in outs]))
        return [o.tobytes()[:n * fr_size] for o in outs]

    def ntt(self, ctx, curve, data: bytes, log_n, inverse=False, coset=False):
        a = np.frombuffer(bytearray(data), dtype=np.uint8)
        self.check(ctx, self.dll.ark355_ntt_fr(ctx, curve, a.ctypes.data_as(C.c_void_p), log_n, int(inverse), int(coset)))
        return a.tobytes()

    def msm(self, ctx, curve, group, bases: bytes, scalars: bytes, n, out_size):
        out = np.zeros(out_size, dtype=np.uint8)
        bb, k1 = _buf(bases if n else None)
        sb, k2 = _buf(scalars if n else None)
        fn = self.dll.ark355_msm_g1 if group == 1 else self.dll.ark355_msm_g2
        self.check(ctx, fn(ctx, curve, bb, sb, n, out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()

    def bases_load(self, ctx, curve, group, bases, n):
        h = C.c_void_p()
        bb, k = _buf(bases if n else None)
        self.check(ctx, self.dll.ark355_bases_load(ctx, curve, group, bb, n, C.byref(h)))
        return h

    def msm_dev(self, ctx, bases_h, d_scalars_ptr, n, mont, out_size, partial=False):
        out = np.zeros(out_size, dtype=np.uint8)
        fn = self.dll.ark355_msm_dev_partial if partial else self.dll.ark355_msm_dev
        self.check(ctx, fn(ctx, bases_h, C.c_void_p(d_scalars_ptr), n, int(mont), out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()

    def xyzz_sum(self, ctx, curve, group, partials: bytes, count, out_size):
        out = np.zeros(out_size, dtype=np.uint8)
        pb, k = _buf(partials if count else None)
        self.check(ctx, self.dll.ark355_xyzz_sum(ctx, curve, group, pb, count, out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()

    def fixed_base_mul(self, ctx, curve, group, base: bytes, scalars, n, point_size):
        out = np.zeros(max(1, n * point_size), dtype=np.uint8)
        bb, k1 = _buf(base)
        sb, k2 = _buf(scalars if n else None)
        self.check(ctx, self.dll.ark355_fixed_base_mul(ctx, curve, group, bb, sb, n, out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()[:n * point_size]

    def setup_scalars(self, curve, n, ell, w, mats, trapdoor: bytes):
        """ark355_setup_scalars: mats as for r1cs_load; trapdoor = tau|alpha|beta|gamma|delta (5 x 32 B canonical).
        Returns dict of canonical byte strings u, v, w (m each), l (w), gamma_abc (ell), h (N - 1)."""
        keep = []
        rp = (C.c_void_p * 3)()
        cl = (C.c_void_p * 3)()
        cf = (C.c_void_p * 3)()
        for i, (row_ptr, col, coeff) in enumerate(mats):
            a = np.ascontiguousarray(row_ptr, dtype=np.uint64)
            b = np.ascontiguousarray(col, dtype=np.uint32)
            if b.size == 0:
                b = np.zeros(1, dtype=np.uint32)
            c, kc = _buf(coeff if len(coeff) else bytes(32))
            keep += [a, b, kc]
            rp[i], cl[i], cf[i] = a.ctypes.data, b.ctypes.data, c.value
        N = 1
        while N < n + ell:
            N <<= 1
        m = ell + w
        outs = {k: np.zeros(max(1, cnt) * 32, dtype=np.uint8)
                for k, cnt in (("u", m), ("v", m), ("w", m), ("l", w), ("gamma_abc", ell), ("h", N - 1))}
        tb, kt = _buf(trapdoor)
        self.check(None, self.dll.ark355_setup_scalars(curve, n, ell, w, C.byref(rp), C.byref(cl), C.byref(cf), tb,
                                                       *[outs[k].ctypes.data_as(C.c_void_p) for k in ("u", "v", "w", "l", "gamma_abc", "h")]))
        cnts = {"u": m, "v": m, "w": m, "l": w, "gamma_abc": ell, "h": N - 1}
        return {k: outs[k][:cnts[k] * 32] for k in outs}

    def verify_batch(self, ctx, curve, vk_parts, proofs, public_inputs: bytes, rho=None) -> bool:
        """ark355_verify_batch.  vk_parts: (alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1) raw images; proofs: list of
        (a, b, c) raw images; public_inputs: count x (ell - 1) Montgomery Fr; rho: list of canonical 32-byte values or None."""
        alpha, beta2, gamma2, delta2, gabc = vk_parts
        keep = []
        d = VkDesc()
        for name, val in (("alpha_g1", alpha), ("beta_g2", beta2), ("gamma_g2", gamma2), ("delta_g2", delta2), ("gamma_abc_g1", gabc)):
            p, k = _buf(val)
            keep.append(k)
            setattr(d, name, p.value)
        g1 = len(alpha)
        d.num_instance = len(gabc) // g1
        arr = (ProofRaw * len(proofs))()
        for i, (a, b, c) in enumerate(proofs):
            C.memmove(arr[i].a, a, len(a))
            C.memmove(arr[i].b, b, len(b))
            C.memmove(arr[i].c, c, len(c))
        ib, k1 = _buf(public_inputs if len(public_inputs) else None)
        rb, k2 = _buf(b"".join(rho) if rho else None)
        ok = C.c_int32(0)
        self.check(ctx, self.dll.ark355_verify_batch(ctx, curve, C.byref(d), arr, ib, rb, len(proofs), C.byref(ok)))
        return bool(ok.value)

    def timings(self, ctx):
        t = Timings()
        self.check(ctx, self.dll.ark355_get_timings(ctx, C.byref(t)))
        return {n: getattr(t, n) for n, _ in Timings._fields_}

    def kernel_stats(self, ctx):
        ms, l, p = C.c_float(0), C.c_uint64(0), C.c_uint64(0)
        self.check(ctx, self.dll.ark355_get_kernel_stats(ctx, C.byref(ms), C.byref(l), C.byref(p)))
        return {"accumulate_ms": ms.value, "launches": l.value, "points": p.value}
