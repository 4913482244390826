#!/usr/bin/env python3
"""bench.py -- Groth16 prove throughput (R1CS constraints/s, BLS12-381) on MI355X.

A "step" is ONE Groth16 proof (witness map: SpMV + 6 NTTs; 4 G1 MSMs + 1 G2 MSM; finalize) of the
workload BASELINE.json's metric is quoted on: configs[1], the 2^20-constraint synthetic R1CS ("S2
mulchain") over BLS12-381, literal n = 2^20 => domain N = 2^21.  Proving key, CSR matrices
and the assignment z are resident in HBM when the timed region starts.

N GPUs, one process per GPU (`--gpus N` spawns the N ranks itself through torch.distributed.run when it is not
already running under a launcher; under the driver's launcher WORLD_SIZE must equal --gpus):
  --mode replica (default)  every rank proves independent instances: the path partitions by independent proofs, no
                            data-path collective, `scaling` "weak", `value` = whole-job aggregate;
  --mode shard              BASELINE configs[2]: ONE proof per step (default n = 2^22), its MSM term ranges sharded
                            over the ranks (ark355_pk_load_shard), the partial sums exchanged by RCCL behind the C ABI
                            (ark355_prove_sharded: all-gather, or --shard-exchange ring for the bucket-level ring
                            reduce-scatter), `scaling` "strong".

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (bucket accumulation) against the
HBM roofline the north star mandates; `cpu_baseline` is the oracle's CPU restatement timed on the same workload on
this box's host cores (the only place the oracle is touched here).

--dry-run-emul (tests only): the same control flow over the CPU emulator build of the library and gloo, tiny n --
checks the launcher / rank / JSON plumbing on a machine without GPUs; its numbers are not measurements and say so.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Secondary, integer roofline of the bucket-accumulation kernels (msm_accumulate28_kernel, msm_accumulate_g2l28_kernel: G1 and
# lane-split G2, two lanes per addition, on radix-2^28 limbs).  The scarce instruction is v_mad_u64_u32.
# Round 6: both numbers are MEASURED, not entered.  The peak is what the chip issues in a 20 ms loop of nothing but that instruction
# at the accumulation kernels' occupancy (ark355_diag_mad_rate, right before and right after the timed region).  A loop that light
# runs at the full 2.4 GHz on every box (31.5-32.5 T/s), so this is the hardware's peak, not the box's sustained rate: what differs
# from box to box is the clock under the REAL load (2.1-2.3 GHz under the 1400 W cap: 4-5 % in every time of this path), and that
# is normalised separately, with the GPU's own cycle counter (`box`, box_block below).  MAD_PEAK_REF_T is the fallback when the
# diagnostic is unavailable (tools/ubench5 on a throttled box, profiles/r05_runA_ubench5.txt).  The multiply-adds per mixed addition are counted in the
# code object of the library that runs (tools/code_object_stats.py at build time -> snark_amd/libark355.stats.json); the table
# below is the fallback for a library built without the stats file and says so in the line.
MAD_PEAK_REF_T = 29.3
MADS_PER_ADD_FALLBACK = {"bls12_381": {"g1": 3155, "g2": 2 * 4410}, "bn254": {"g1": 1526, "g2": 2 * 2162}}


def mads_per_add(curve):
    """(table, source): v_mad_u64_u32 + v_mad_i64_i32 per mixed addition of the shipped accumulation kernels (G2: both lanes)."""
    try:
        import snark_amd
        st = json.load(open(os.path.splitext(snark_amd.LIB_PATH)[0] + ".stats.json"))
        g1 = st[curve + ".g1"]["hot_block"]["multiply_adds"]
        g2 = st[curve + ".g2"]["hot_block"]["multiply_adds"]
        if g1 > 0 and g2 > 0:
            return {"g1": g1, "g2": 2 * g2}, "code object of the loaded library (tools/code_object_stats.py, hot path of the kernels)"
    except Exception:                                         # noqa: BLE001
        pass
    return MADS_PER_ADD_FALLBACK[curve], "fallback constants (no libark355.stats.json next to the library)"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=None,
                    help="log2 of the constraint count (default: 20 = BASELINE configs[1]; 22 = configs[2] in --mode shard)")
    ap.add_argument("--mode", default="replica", choices=["replica", "shard"])
    ap.add_argument("--shard-exchange", default="window", choices=["window", "ring"])
    ap.add_argument("--dry-run-emul", action="store_true", help="tests only: CPU emulator + gloo, not a measurement")
    ap.add_argument("--curve", default="bls12_381", choices=["bls12_381", "bn254"])
    ap.add_argument("--tight", action="store_true", help="domain-tight variant n = 2^k - 100 (N = 2^k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the closed-form proof check")
    ap.add_argument("--inflight", type=int, default=4,
                    help="proofs in flight per GPU (independent contexts sharing the resident key; 1 = strictly serial)")
    ap.add_argument("--no-ab", action="store_true", help="skip the in-run A/B of the schedules (after the timed region)")
    ap.add_argument("--no-micro", action="store_true", help="skip the stand-alone MSM / NTT readings (after the timed region)")
    ap.add_argument("--no-telemetry", action="store_true", help="no clock / power sampling")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the synthesis-in-the-loop reading (host mirror feeding in-flight proofs; after the timed region)")
    ap.add_argument("--e2e-s3", action="store_true",
                    help="the e2e block also runs the S3 bench-LC circuit (mirror-bound: ~50 s, says nothing about a Rust host)")
    ap.add_argument("--profile-run", action="store_true",
                    help="for rocprofv3: nothing but preparation, warm-up and the timed region (no A/B, isolated, single-proof, "
                         "latency or stand-alone readings), so that every proof of the trace ran under ONE schedule (ARK355_SCHED)")
    args = ap.parse_args()
    if args.profile_run:
        args.no_ab = args.no_micro = args.no_telemetry = args.no_cpu_baseline = args.no_e2e = True
    if args.dry_run_emul:
        args.inflight = 1                # the emulator is single-threaded
    if args.log_n is None:
        args.log_n = (8 if args.dry_run_emul else 22) if args.mode == "shard" else (6 if args.dry_run_emul else 20)
    return args


def spawn_ranks(args):
    """`bench.py --gpus N` outside a launcher: start the N ranks (one per GPU) and relay rank 0's JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(curve_name, log_n_sample=None, workload=None):
    """Oracle CPU restatement on a bounded sample of the same workload (rank 0, N=1 only).  workload: the bench's own statement,
    key and one of its GPU proofs -- on a host that can time the full size the CPU proofs run on exactly that, and the first one
    is compared byte for byte with the GPU's (`parity_vs_oracle`): the oracle in its role as checker."""
    try:
        from oracle.c import cbase
        return cbase.bench_prove(curve_name, log_n_sample, workload=workload)
    except Exception as e:            # pragma: no cover - fallback keeps the bench line complete
        sys.stderr.write("[bench] C oracle unavailable (%r); timing the Python oracle on a tiny sample\n" % (e,))
    from oracle import groth16 as G, synthetic as S
    from oracle.fields import CURVES
    C = CURVES[curve_name]
    n = 1 << 7
    A, B, Cm, z, ell = S.mulchain_direct(C.r, n)
    pk = G.setup(C, A, B, Cm, ell, len(z), G.Trapdoor(12345, 2, 3, 4, 5))
    t0 = time.perf_counter()
    G.prove(C, pk, A, B, Cm, z, ell, 7, 9)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "constraints/s", "cores": 1, "kind": "port",
            "sample": "pure-Python oracle (scalar, naive MSM), 2^7-constraint mulchain, 1 proof"}


def e2e_reading(curve, n, inflight, device_value, s3=False):
    """SURVEY 8f-4, outside the timed region: SNARK::prove WITH synthesis in the loop.  K witness-only synthesis threads (one
    constraint system each -- the reference's ConstraintSystemRef is Rc<RefCell>, so a thread per proof is its parallel unit)
    hand assignments in page-locked buffers to `inflight` proving threads (Groth16::prove_pipelined of host_mirror/snark.hpp,
    the C++ stand-in for a Rust host; tests/cpp/test_host_mirror --e2e), in its own process with its own key.  The circuit is S2,
    the bench's own mulchain (one field multiplication of host work per constraint); `value` of the block is that reading.
    s3=True (flag --e2e-s3) adds S3, the reference's benchmark SHAPE (relations/examples/bench.rs:22-83: up to ten terms per
    linear combination) made satisfiable by evaluating those combinations in the witness closures -- ~45 field multiplications
    per constraint that the reference's own closures (`Ok(self.a)`, bench.rs:74-76) do not do, on a C++ mirror that is not
    ark-relations: the S3 figure is MIRROR-BOUND and says nothing about a Rust host; it is labelled so."""
    import subprocess
    try:
        from snark_amd import build as B
        exe = B.build_host_mirror_exe()
        quota = os.cpu_count() or 1
        try:                                       # the container's CPU quota, not the host's thread count
            q = open("/sys/fs/cgroup/cpu.max").read().split()
            if q[0] != "max":
                quota = max(1, int(int(q[0]) / int(q[1])))
        except (OSError, ValueError, IndexError):
            pass
        threads = max(2, min(12, quota - 2 - inflight // 2))
        env = dict(os.environ)
        env.pop("ARK355_E2E_SWEEP", None)
        out = {"unit": "constraints/s", "synthesis_threads": threads, "inflight": inflight, "host_cpu_quota_cores": quota,
               "host": "C++ mirror of ark-relations (host_mirror/), witness-only synthesis; a Rust host runs the real crate"}
        cases = [("s2_mulchain", "mulchain", 6 * inflight, [])]
        if s3:
            cases.append(("s3_bench_lc_mirror_bound", "benchlc", max(inflight, threads), ["e2e-only"]))
        for tag, circuit, count, extra in cases:
            t0 = time.perf_counter()
            try:
                r = subprocess.run([exe, "--e2e", curve, str(n), str(count), str(threads), str(inflight), circuit] + extra,
                                   capture_output=True, text=True, timeout=150, env=env)
            except subprocess.TimeoutExpired:
                out[tag] = {"error": "timed out after 150 s"}
                continue
            kv = dict(l.split("=", 1) for l in r.stdout.splitlines() if "=" in l and not l.startswith("sweep"))
            if r.returncode != 0 or "e2e_constraints_per_s" not in kv:
                out[tag] = {"error": "rc=%d %s" % (r.returncode, (r.stderr or r.stdout)[-200:])}
                continue
            val, wall = float(kv["e2e_constraints_per_s"]), float(kv["e2e_wall_s"])
            rec = {"value": val, "proofs": count, "n": n, "ratio_to_value": val / device_value if device_value else None,
                   "synthesis_cpu_cores_used": float(kv["e2e_synth_cpu_s"]) / wall if wall > 0 else None,
                   "seconds": round(time.perf_counter() - t0, 2)}
            dev = float(kv.get("device_only_pinned_constraints_per_s", 0))
            if dev:
                rec["device_only_same_process"] = dev
                rec["ratio_to_device_only_same_process"] = val / dev
            best = max(rec["ratio_to_value"] or 0, rec.get("ratio_to_device_only_same_process") or 0)
            rec["bound"] = "device" if best >= 0.9 else "host synthesis (%d threads of a %d-core quota)" % (threads, quota)
            out[tag] = rec
        if "value" in out.get("s2_mulchain", {}):
            out["value"] = out["s2_mulchain"]["value"]
        if "s3_bench_lc_mirror_bound" in out and "value" in out["s3_bench_lc_mirror_bound"]:
            out["s3_bench_lc_mirror_bound"]["note"] = ("bounded by the C++ mirror's witness closures (they evaluate the linear combinations; the "
                                                       "reference's bench.rs closures return a stored value), not by the path")
        return out
    except Exception as e:                                    # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def thread_cpu_times():
    """CPU seconds (user + system) of every thread of this process, keyed by (tid, comm): where the host cores of a rank go."""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                st = open("/proc/self/task/%s/stat" % tid).read()
            except OSError:
                continue
            comm = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[(tid, comm)] = (int(f[11]) + int(f[12])) / tick
    except OSError:
        pass
    return out


REF_GFXCLK_MHZ = 2200.0       # the clock the normalised figures are quoted at (boxes of the pool sustain 2.1-2.3 GHz under the 1400 W cap)


def box_block(gpu_clocks, dt, steps, value, n, telemetry):
    """Box-independent reading of the timed region.  The GPU counts its own shader-clock cycles (s_memtime) and a constant 100 MHz
    reference (s_memrealtime); read right before and right after the region (ark355_diag_clocks) they give the cycles the region
    took at whatever clock THIS box sustained under its power cap: cycles per constraint is what to compare across boxes and
    commits, and `*_at_ref_clock` are the headline figures rescaled to REF_GFXCLK_MHZ."""
    out = {"ref_gfxclk_mhz": REF_GFXCLK_MHZ}
    try:
        from snark_amd._binding import Lib
        clk, ms, units = Lib.diag_clocks_delta(gpu_clocks["t0"], gpu_clocks["t1"])
        if clk:
            cycles = clk * 1e3 * ms                            # MHz x ms = 10^3 cycles
            out["gfxclk_mhz_mean_on_chip"] = clk
            out["compute_units_read"] = units
            out["region_ms_on_chip"] = ms
            out["gfx_cycles_per_step"] = cycles / steps
            out["gfx_cycles_per_constraint"] = cycles / steps / n
            out["ms_per_step_at_ref_clock"] = cycles / steps / (REF_GFXCLK_MHZ * 1e3)
            out["value_at_ref_clock"] = value * (REF_GFXCLK_MHZ / clk)
    except Exception as e:                                    # noqa: BLE001
        out["error"] = ("no on-chip counters: %s %s" % (type(e).__name__, {k: v for k, v in gpu_clocks.items() if k.endswith("_error")}))[:200]
    try:
        out["gfxclk_mhz_mean_smi"] = telemetry["timed_region"]["current_gfxclk"]["mean"]
    except Exception:                                         # noqa: BLE001
        pass
    out["note"] = ("cycles = the GPU's shader-clock counter over the timed region (host launch gaps included: they are part of the step); "
                   "compare gfx_cycles_per_constraint / *_at_ref_clock across boxes and commits, raw ms_per_step / value only within one box")
    return out


def pmc_traffic(n, curve):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE, separate runs; profiles/README.md).  profiles/pmc_latest.json holds one record per workload
    ("workloads": {"<curve>:n=<constraints>": {...}}; the single-record layout of rounds 3-4 is still read).  None when
    no profile of this workload is on file."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        top = json.load(open(path))
        key = "%s:n=%d" % (curve, n)
        rec = (top.get("workloads") or {}).get(key)
        if rec is None and top.get("workload") == key:
            rec = top
        if rec is None:
            return None, None
        k = rec["msm_accumulate_kernel"]
        note = ("static: FETCH_SIZE + WRITE_SIZE of rocprofv3 --pmc passes recorded in profiles/pmc_latest.json (%s), "
                "not collected by this run" % rec.get("recorded", "date not recorded"))
        return k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"], note
    except Exception:
        return None, None


def micro_readings(g, cv, pk, torch, np, dev_sync, rnd):
    """Stand-alone MSM (ark355_bases_load + ark355_msm_dev) and lone-NTT (ark355_ntt_fr_dev) latencies on this GPU."""
    import ctypes
    L, ctx, cid = g.lib, g.ctx, cv.curve_id
    s1, s2 = g.sizes["g1"], g.sizes["g2"]

    def scalars(n, dist, gen):
        top = (1 << (cv.r.bit_length() - 1 - 192)) - 1
        uni = gen.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64, endpoint=False)
        uni[:, 3] &= np.uint64(top)
        if dist == "uniform":
            return uni
        if dist == "equal":
            return np.tile(uni[:1], (n, 1))
        raw = np.zeros((n, 4), dtype="<u8")
        kind = gen.integers(0, 256, size=n, dtype=np.uint8)
        raw[:, 0] = kind & 1
        sel = kind >= 230
        raw[sel] = uni[sel]
        return raw

    gen = np.random.default_rng(rnd.getrandbits(63))
    out = {"definition": "latency of ONE ark355_msm_dev call over resident window tables (digit sort, bucket accumulation, "
                         "merge / reduction / combination, normalisation, D2H of the point), median of 5; scalars canonical, "
                         "uniform / all equal / 90 % boolean; bases = the key's own query vectors",
           "msm": {}}

    def as_u8(b):
        return np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b.view(np.uint8).reshape(-1)
    a_q, b1_q, b2_q, h_q, l_q = (as_u8(x) for x in (pk.a_query, pk.b_g1_query, pk.b_g2_query, pk.h_query, pk.l_query))
    n20 = min(len(a_q) // s1, 1 << 20)
    cases = [("g1", 1, a_q[:n20 * s1], n20, ("uniform", "equal", "boolean")),
             ("g2", 2, b2_q[:n20 * s2], n20, ("uniform", "equal", "boolean"))]
    big = np.concatenate([a_q, b1_q, l_q, h_q])
    n22 = min(len(big) // s1, 1 << 22)
    if n22 > n20:
        cases.append(("g1", 1, big[:n22 * s1], n22, ("uniform",)))
    for name, group, bases, n, dists in cases:
        psz = s1 if group == 1 else s2
        h = L.bases_load(ctx, cid, group, np.ascontiguousarray(bases), n)
        try:
            for dist in dists:
                ks = scalars(n, dist, gen)
                kd = torch.from_numpy(ks.view(np.uint8).reshape(-1).copy()).cuda()
                dev_sync()
                L.msm_dev(ctx, h, kd.data_ptr(), n, 0, psz)
                ts = []
                for _ in range(5):
                    ta = time.perf_counter()
                    L.msm_dev(ctx, h, kd.data_ptr(), n, 0, psz)
                    ts.append((time.perf_counter() - ta) * 1e3)
                ms = sorted(ts)[2]
                out["msm"]["%s_2^%d_%s" % (name, n.bit_length() - 1, dist)] = {
                    "n": n, "ms": round(ms, 3), "mscalar_mul_per_s": round(n / ms / 1e3, 1),
                    "accumulate_kernel_ms": round(L.kernel_stats(ctx)["accumulate_ms"], 3)}
                del kd
        finally:
            L.dll.ark355_bases_free(h)
    # lone forward NTT, 2^21 points, data and scratch resident (ark355_ntt_fr_dev on the context's stream)
    log_n = 21
    nb = (1 << log_n) * 32
    data = torch.from_numpy(gen.integers(0, 1 << 62, size=(1 << log_n) * 4, dtype=np.uint64).view(np.uint8).copy()).cuda()
    scr = torch.empty(nb, dtype=torch.uint8, device=data.device)
    dev_sync()
    fn = L.dll.ark355_ntt_fr_dev

    def ntt_once():
        rc = fn(ctx, cid, ctypes.c_void_p(data.data_ptr()), ctypes.c_void_p(scr.data_ptr()), log_n, 0, 0, None)
        assert rc == 0, rc
    ntt_once()
    dev_sync()
    reps = 20
    ta = time.perf_counter()
    for _ in range(reps):
        ntt_once()
    dev_sync()
    ms = (time.perf_counter() - ta) / reps * 1e3
    out["ntt"] = {"log_n": log_n, "ms": round(ms, 4), "elements_per_s": (1 << log_n) / (ms * 1e-3),
                  "definition": "forward radix-2 NTT of 2^21 Fr elements resident in HBM, %d back-to-back calls" % reps}
    return out


_T0 = time.perf_counter()
_JSON_FD = None


def stage(msg):
    """progress marker on stderr (stdout carries the one JSON line only)"""
    sys.stderr.write("[bench %7.1f s] %s\n" % (time.perf_counter() - _T0, msg))
    sys.stderr.flush()


def main():
    args = parse_args()
    # stdout carries ONE line, the JSON record: whatever the libraries underneath print there (RCCL announces its version on
    # stdout when a communicator is created) goes to stderr instead, and the record is written to the saved descriptor
    global _JSON_FD
    if _JSON_FD is None and not (args.gpus > 1 and "WORLD_SIZE" not in os.environ):
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)
    wd = os.environ.get("ARK355_BENCH_WATCHDOG")
    if wd:                                   # diagnostic: dump every thread's Python stack and exit if the run takes longer
        import faulthandler
        faulthandler.dump_traceback_later(float(wd), exit=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but the launcher started %d ranks" % (args.gpus, world)
    import numpy as np
    import torch
    import torch.distributed as dist
    emul = args.dry_run_emul
    if emul:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
        import build_emul
        from snark_amd._binding import Lib
        backend_lib = Lib(build_emul.build())
        dev_sync = lambda: None                                  # noqa: E731
        xdev = "cpu"
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
        torch.cuda.set_device(local_rank)
        backend_lib = None                                       # snark_amd.lib(): libark355.so or a loud failure
        dev_sync = torch.cuda.synchronize
        xdev = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emul:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # one collective over all ranks before anything else: the N ranks (and the RCCL ring over xGMI) exist
        probe = torch.ones(1, dtype=torch.int64, device=xdev)
        dist.all_reduce(probe)
        assert int(probe.item()) == world
    from snark_amd import params, synthetic
    from snark_amd.groth16 import Groth16
    import random

    shard = args.mode == "shard"
    stage("start: rank %d of %d" % (rank, world))
    cv = params.CURVES[args.curve]
    n = (1 << args.log_n) - (100 if args.tight else 0)
    t_prep = time.perf_counter()
    dev_id = 0 if emul else local_rank                           # the emulator models one device
    g = Groth16(cv, device=dev_id, lib=backend_lib)
    # replica mode: independent instances per rank; shard mode: every rank holds the SAME statement and key
    seed = 0x355 + (0 if shard else rank)
    r1, z = synthetic.mulchain(cv, n, seed=seed)
    rnd = random.Random(seed)
    stage("statement built (n = %d)" % n)
    pk, vk = g.circuit_specific_setup(r1, lambda: rnd.randrange(1, cv.r), keep_trapdoor=not args.no_check)
    stage("key generated")
    sg = None
    if shard:
        from snark_amd.parallel import ShardedGroth16, SHARD_BUCKET_RING, SHARD_WINDOW
        if world == 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("gloo" if emul else "nccl", rank=0, world_size=1)
        sg = ShardedGroth16(g, device=xdev)
        sg.load_pk_shard(pk)
        shard_mode = SHARD_BUCKET_RING if args.shard_exchange == "ring" else SHARD_WINDOW
        args.inflight = 1                # one collective proof at a time
    else:
        g.load_pk(pk)
    g.load_r1cs(r1)
    zb = synthetic.z_to_mont_bytes(cv, z)
    if emul:
        z_host = np.frombuffer(zb, dtype=np.uint8).copy()

        class _Z:                                                # "device" memory of the emulator is host memory
            def data_ptr(self):
                return z_host.ctypes.data
        z_dev = _Z()
    else:
        z_dev = torch.from_numpy(np.frombuffer(zb, dtype=np.uint8).copy()).cuda()     # z resident in HBM
    dev_sync()
    prep_s = time.perf_counter() - t_prep
    stage("key, matrices and z resident")

    # Throughput mode: `inflight` independent proving contexts (own streams and scratch) share the resident key and
    # CSR handles; while one proof is in its serial head (sort) or tail (last bucket reduction, O(1) host finish)
    # the other keeps the CUs busy.  Every step is still one complete, independently randomised proof.
    import threading
    from snark_amd.groth16 import Proof
    rh = g.load_r1cs(r1)
    pkh = None if shard else g.load_pk(pk)
    ctxs = [g.ctx] + [g.lib.ctx_create(dev_id) for _ in range(max(1, args.inflight) - 1)]
    lock = threading.Lock()

    def prove_on(ctx, r_, s_, z_host=None):
        if shard:
            if z_host is not None:
                return sg.prove(pk, r1, z_host, r_, s_, mode=shard_mode)
            return sg.prove(pk, r1, None, r_, s_, mode=shard_mode, z_device_ptr=z_dev.data_ptr())
        if z_host is not None:           # SURVEY 8d window: assignment in HOST memory -> proof in host memory (H2D included)
            a, b, c = g.lib.prove(ctx, pkh, rh, z_host, r1.m, cv.fr_canon(r_), cv.fr_canon(s_), g.sizes)
        else:
            a, b, c = g.lib.prove(ctx, pkh, rh, z_dev.data_ptr(), r1.m, cv.fr_canon(r_), cv.fr_canon(s_), g.sizes,
                                  z_is_device_ptr=True)
        return Proof(a, b, c)

    trace = [] if os.environ.get("ARK355_BENCH_TRACE") else None
    worker_cpu = [0.0]                   # CPU seconds of the proving threads (they are created and joined per run())

    def run(nsteps, record, per_worker=False, z_host=None):
        # per_worker: every context proves `nsteps` times (warm-up must touch each context's scratch and streams)
        todo = [(rnd.randrange(cv.r), rnd.randrange(cv.r)) for _ in range(nsteps * (len(ctxs) if per_worker else 1))]
        quota = {id(c): nsteps for c in ctxs}
        results = []

        def worker(ctx):
            cpu_a = time.thread_time()
            while True:
                with lock:
                    if not todo or (per_worker and quota[id(ctx)] == 0):
                        worker_cpu[0] += time.thread_time() - cpu_a      # the thread is gone when /proc is read
                        return
                    quota[id(ctx)] -= 1
                    r_, s_ = todo.pop()
                t_a = time.perf_counter()
                p = prove_on(ctx, r_, s_, z_host)  # the C call releases the GIL; it returns after its streams drained
                t_b = time.perf_counter()
                ks = g.lib.kernel_stats(ctx)
                if trace is not None:
                    trace.append((t_a, t_b))
                with lock:
                    results.append((p, r_, s_))
                    if record is not None:
                        record[0] += ks["accumulate_ms"]
                        record[1] += ks["launches"]
                        record[2] += ks["points"]
        ths = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return results

    # ---- untimed preparation that belongs to the library's own start-up ------------------------------------------
    # telemetry (tools/gpu_telemetry.py): static facts now, clocks / power / throttling sampled across the run
    telemetry = {"note": "tools/gpu_telemetry.py; sampled by a child process every 25 ms"}
    sampler = None
    if rank == 0 and not emul and not args.no_telemetry:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import gpu_telemetry as GT
            bus = None
            try:
                pr = torch.cuda.get_device_properties(local_rank)
                bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
                telemetry["device"] = {"name": pr.name, "cus": pr.multi_processor_count, "hbm_bytes": pr.total_memory,
                                       "torch": torch.__version__, "hip": torch.version.hip}
            except Exception as e:                            # noqa: BLE001
                telemetry["device"] = {"error": str(e)[:120]}
            telemetry["static"] = GT.snapshot(bus)
            sampler = GT.Sampler(bus, period_ms=25)
            telemetry["host_before"] = GT.host_counters()
        except Exception as e:                                # noqa: BLE001
            telemetry["error"] = "%s: %s" % (type(e).__name__, str(e)[:160])
    # which of the proving streams share an in-order hardware queue of the runtime (idle device; see ark355_diag_streams)
    stream_map = None
    if not emul and rank == 0 and not args.profile_run:
        try:
            # GPU-side cost of a dispatch in an in-order stream on THIS box (ark355_diag_dispatch): the fingerprint that
            # separates the boxes of the pool -- a few us on most, 50-90 us on some, where everything but the long
            # accumulation kernels runs 2-3x slower
            telemetry["dispatch_gap_before"] = g.lib.diag_dispatch(g.ctx)
        except Exception as e:                                # noqa: BLE001
            telemetry["dispatch_gap_before"] = {"error": str(e)[:160]}
        try:
            mat = g.lib.diag_streams(ctxs)
            names = ["ctx%d" % i for i in range(len(ctxs))] + ["ctx0.sW", "ctx0.sS", "ctx0.sR"]
            pairs = [[names[i], names[j]] for i in range(len(names)) for j in range(i + 1, len(names))
                     if mat[i][j] == 1 or mat[j][i] == 1]
            stream_map = {"streams": names, "serialised_pairs": pairs,
                          "note": "pairs of streams on which a kernel waited for a spinning kernel of the other: they "
                                  "share a hardware queue (GPU_MAX_HW_QUEUES) and cannot overlap"}
        except Exception as e:                                # noqa: BLE001
            stream_map = {"error": str(e)[:160]}
    # schedule calibration: with policy SCHED = AUTO the library tries its candidate schedules on the first warm proofs of
    # a class and keeps the fastest; let it finish BEFORE the contract's warm-up so that the timed region runs one schedule
    calib = None
    if not shard and pkh is not None and not emul:
        t_c = time.perf_counter()
        rounds = 0
        # the in-flight class explores in four phases of 8 + 16 completions: ONE continuous run of proofs (the ramp-down and
        # ramp-up between separate runs would sit inside the scored windows), then shorter runs until it has latched
        per_round = max(1, -(-(96 + 4 * len(ctxs)) // len(ctxs)))      # four phases of 24 completions + the stragglers between them
        while rounds < 4:
            run(per_round if rounds == 0 else max(1, per_round // 4), None, per_worker=True)
            rounds += 1
            info = g.lib.sched_info(g.ctx, pkh, len(ctxs) > 1)
            if info["latched"] != "auto" or g.lib.ctx_get_policy(g.ctx, "SCHED") >= 0:
                break
        calib = {"proofs": (per_round + (rounds - 1) * max(1, per_round // 4)) * len(ctxs), "seconds": round(time.perf_counter() - t_c, 3),
                 "in_flight" if len(ctxs) > 1 else "alone": g.lib.sched_info(g.ctx, pkh, len(ctxs) > 1)}
        stage("schedule calibration done: %s" % json.dumps(calib))
    run(max(1, -(-args.warmup // len(ctxs))) if args.warmup else 0, None, per_worker=True)
    stage("warm-up done")
    # The harness keeps the assignment as a list of 2^20 Python ints (for the closed-form check) next to other large
    # containers; a full cyclic-GC pass over them costs ~24 ms and fired once per timed region (seen with
    # ARK355_BENCH_TRACE=1: one call of 56 ms among calls of 32 ms, while the library's own timer showed 32 ms for all).
    # Park everything allocated so far in the permanent generation: the collector no longer walks it.
    import gc
    gc.collect()
    gc.freeze()
    rec = [0.0, 0, 0]
    # the multiply-add rate of this box, warm, right before (and right after) the timed region -- outside it
    mad_peak = {}
    def read_mad_peak(tag):
        if emul or rank != 0 or args.profile_run:
            return
        try:
            mad_peak[tag] = g.lib.diag_mad_rate(g.ctx, 20.0)["tmad_per_s"]
        except Exception as e:                                # noqa: BLE001
            mad_peak[tag + "_error"] = str(e)[:120]
    read_mad_peak("before")
    gpu_clocks = {}
    def read_gpu_clocks(tag):
        # shader-clock cycles and 100 MHz ticks of the GPU's own counters (ark355_diag_clocks), bracketing the timed region
        if emul or rank != 0 or args.profile_run:
            return
        try:
            gpu_clocks[tag] = g.lib.diag_clocks(g.ctx)
        except Exception as e:                                # noqa: BLE001
            gpu_clocks[tag + "_error"] = str(e)[:120]
    if world > 1:
        dist.barrier()
    dev_sync()
    read_gpu_clocks("t0")
    thr0 = thread_cpu_times()
    worker_cpu[0] = 0.0
    cpu0 = time.process_time()
    wall0 = time.time()
    t0 = time.perf_counter()
    results = run(args.steps, rec)
    dev_sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    wall1 = time.time()
    read_gpu_clocks("t1")
    read_mad_peak("after")
    if sampler is not None:
        time.sleep(0.06)                         # let the sampler's last reading of the region land
        telemetry["timed_region"] = sampler.window(wall0, wall1)
        telemetry["host_after_timed_region"] = GT.host_counters()
    host_cpu_s = time.process_time() - cpu0          # CPU time of all threads of this rank over the timed region
    thr1 = thread_cpu_times()
    stage("timed region done: %.2f ms per step" % (dt / args.steps * 1e3))
    resident = sum(max(0.0, t1_ - thr0.get(key, 0.0)) for key, t1_ in thr1.items())
    # where the cores go: the proving threads (launches, event polling, result copies), the threads that live through the
    # region (main thread, the HIP runtime's own), and what is left: short-lived native threads (the O(1) proof tail)
    host_cpu_threads = {"proving_threads": round(worker_cpu[0] / dt, 3), "resident_threads": round(resident / dt, 3),
                        "short_lived_native_threads": round(max(0.0, host_cpu_s - worker_cpu[0] - resident) / dt, 3)} if dt > 0 else {}
    if trace is not None and rank == 0:
        tl = sorted(trace)[-args.steps:]
        sys.stderr.write("[bench] timed region: t0 -> first call %.2f ms; calls (start, duration ms): %s; last return -> end %.2f ms\n" % (
            (tl[0][0] - t0) * 1e3, ["%.1f+%.1f" % ((a - t0) * 1e3, (b - a) * 1e3) for a, b in tl],
            (t0 + dt - max(b for _, b in tl)) * 1e3))
    acc_ms_sum, acc_launches, acc_points = rec
    last = results[-1]
    tim = g.lib.timings(g.ctx)
    sched_used = None if shard or pkh is None else g.lib.sched_info(g.ctx, pkh, len(ctxs) > 1)
    # In-run A/B of the schedules (outside the timed region; the headline above is the library default): the timed region
    # repeated under each forced schedule, two interleaved passes, same process, same box, same minute.
    ab = None
    if not shard and not args.no_ab and rank == 0 and world == 1 and not emul:
        names = {0: "one_stream", 1: "pipeline", 2: "pipeline_sync", 3: "one_stream_spin"}
        ab = {"steps": args.steps, "inflight": len(ctxs), "ms_per_step": {v: [] for v in names.values()},
              "note": "the timed region repeated under each forced schedule (policy SCHED), interleaved; headline = library default"}
        wall_ab0 = time.time()
        for _pass in range(2):
            for code, nm in names.items():
                for c in ctxs:
                    g.lib.ctx_set_policy(c, "SCHED", code)
                dev_sync()
                ta = time.perf_counter()
                res = run(args.steps, None)
                dev_sync()
                ab["ms_per_step"][nm].append(round((time.perf_counter() - ta) / args.steps * 1e3, 3))
                results.extend(res)
        for c in ctxs:
            g.lib.ctx_set_policy(c, "SCHED", -1)
        if sampler is not None:
            telemetry["ab_block"] = sampler.window(wall_ab0, time.time())
        stage("schedule A/B done: %s" % json.dumps(ab["ms_per_step"]))
    # One proof at a time on ONE stream, nothing else on the device: every kernel runs alone, so the event-bracketed phases
    # are ISOLATED kernel times -- the box-independent check of the kernels themselves ("kernels equal, overlap slower" shows
    # at a glance), and the clean source of the secondary metrics (MSM scalar-mul/s, NTT elements/s).
    isolated = None
    if not shard and not emul and rank == 0 and not args.profile_run:
        g.lib.ctx_set_policy(g.ctx, "SCHED", 0)
        # (a lone one-stream proof normally runs its witness map and its G2 tails on side streams: not here -- ONE stream)
        side = {k: g.lib.ctx_get_policy(g.ctx, k) for k in ("SIDE_WM", "SIDE_G2_TAILS")}
        for k in side:
            g.lib.ctx_set_policy(g.ctx, k, 0)
        acc, tms = [], []
        for _ in range(3):
            prove_on(g.ctx, rnd.randrange(cv.r), rnd.randrange(cv.r))
            acc.append(g.lib.kernel_stats(g.ctx)["accumulate_ms"])
            tms.append(g.lib.timings(g.ctx))
        g.lib.ctx_set_policy(g.ctx, "SCHED", -1)
        for k, v in side.items():
            g.lib.ctx_set_policy(g.ctx, k, v)
        dev_sync()
        tmed = {k: sorted(t[k] for t in tms)[1] for k in tms[0]}
        N_ = r1.domain_size
        isolated = {"definition": "one proof alone on one stream (policy SCHED = 0): every kernel runs by itself; median of 3",
                    "accumulate_ms_per_proof": sorted(acc)[1],
                    "accumulate_g2_launch_ms": tmed["msm_b_g2_ms"], "accumulate_h_launch_ms": tmed["msm_h_ms"],
                    "witness_map_ms": tmed["witness_map_ms"], "total_ms": tmed["total_ms"],
                    "msm_g1_mscalar_mul_per_s": (N_ - 1) / (tmed["msm_h_ms"] * 1e-3) / 1e6 if tmed["msm_h_ms"] > 0 else None,
                    "msm_g2_mscalar_mul_per_s": (r1.m + 4) / (tmed["msm_b_g2_ms"] * 1e-3) / 1e6 if tmed["msm_b_g2_ms"] > 0 else None,
                    "ntt_elements_per_s_witness_map": 6 * N_ / (tmed["witness_map_ms"] * 1e-3) if tmed["witness_map_ms"] > 0 else None,
                    "note": "MSM rates: bucket accumulation + its merge / reduction / combination of the H (G1, N-1 terms) and "
                            "B2 (G2, m+4 terms) MSMs of a proof, sort excluded; NTT rate: the 6 transforms of N points the library's witness map runs "
                            "(the reference's algorithm: 7), SpMV and the elementwise steps included in the time"}
        stage("isolated single-stream reading done")
    # Outside the timed region: single proofs on ONE context, nothing else in flight -- first until the library's measured
    # schedule choice for the "alone" class has latched, then three for the uncontended launch duration of the dominant
    # kernel (with several proofs in flight its launches share the chip with the other proofs' kernels)
    solo = None
    if not shard and pkh is not None and not emul and not args.profile_run:
        for _ in range(16):
            if g.lib.sched_info(g.ctx, pkh, False)["latched"] != "auto":
                break
            prove_on(g.ctx, rnd.randrange(cv.r), rnd.randrange(cv.r))
    if not shard and len(ctxs) > 1 and not args.profile_run:
        solo_rec = [0.0, 0, 0]
        wm_side = g.lib.ctx_get_policy(g.ctx, "SIDE_WM")        # the accumulation launches ALONE: no witness map beside the first ones
        g.lib.ctx_set_policy(g.ctx, "SIDE_WM", 0)
        for _ in range(3):
            prove_on(g.ctx, rnd.randrange(cv.r), rnd.randrange(cv.r))
            ks = g.lib.kernel_stats(g.ctx)
            solo_rec[0] += ks["accumulate_ms"]
            solo_rec[1] += ks["launches"]
        g.lib.ctx_set_policy(g.ctx, "SIDE_WM", wm_side)
        dev_sync()
        solo = solo_rec[0] / max(1, solo_rec[1])
        tim = g.lib.timings(g.ctx)
    # SURVEY.md 8d / BASELINE.md section 3: t_prove = assignment resident in HOST memory -> three affine points in host
    # memory, median of >= 5 single proofs, H2D of z included.  Outside the timed region, rank 0's GPU only; page-locked
    # (ark355_host_alloc) and pageable host buffers, next to the device-resident reading of the same loop; then the
    # in-flight throughput of the timed region repeated from host buffers.
    stage("single-stream reading done")
    latency = None
    # (a sharded proof is a collective: with several ranks rank 0 cannot prove on its own -- there every step already IS a
    # single proof from start to finish, so ms_per_step is the latency)
    if rank == 0 and not (shard and world > 1) and not args.profile_run and (not emul or os.environ.get('ARK355_BENCH_EMUL_LATENCY')):
        import ctypes
        ptr = ctypes.c_void_p()
        assert g.lib.dll.ark355_host_alloc(len(zb), ctypes.byref(ptr)) == 0
        z_pinned = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(len(zb),))
        z_pinned[:] = np.frombuffer(zb, dtype=np.uint8)
        z_page = np.frombuffer(zb, dtype=np.uint8).copy()

        def med(z_host, reps=7):
            ts = []
            for _ in range(reps):
                ta = time.perf_counter()
                prove_on(g.ctx, rnd.randrange(cv.r), rnd.randrange(cv.r), z_host)
                ts.append((time.perf_counter() - ta) * 1e3)
            ts.sort()
            return ts[len(ts) // 2]

        def thr(z_host):
            dev_sync()
            ta = time.perf_counter()
            res = run(args.steps, None, z_host=z_host)
            dev_sync()
            tb = time.perf_counter() - ta
            results.extend(res)                       # checked against the closed form with the others
            return {"ms_per_step": tb / args.steps * 1e3, "value": n * args.steps / tb}

        latency = {"definition": "one proof at a time, wall clock of the C call: assignment z in host memory -> proof "
                                 "(3 affine points) in host memory, H2D of z included; median of 7",
                   "host_pinned_z_ms": med(z_pinned), "host_pageable_z_ms": med(z_page), "device_z_ms": med(None),
                   "inflight_from_host_z": {"pinned": thr(z_pinned), "pageable": thr(z_page),
                                            "note": "the timed region repeated with every proof reading z from host memory"}}
        latency["constraints_per_s_single_proof_host_pinned_z"] = n / (latency["host_pinned_z_ms"] * 1e-3)
        g.lib.dll.ark355_host_free(ptr)
        stage("latency / host-z readings done")
    # Stand-alone readings of BASELINE's secondary metrics (outside the timed region, rank 0): ark355_msm_dev over resident
    # window tables (latency of one call: digit sort, accumulation, bucket reduction, normalisation, D2H of the result) for
    # three scalar distributions, and a lone forward NTT on device-resident data.  Bases: the key's own query vectors.
    micro = None
    if rank == 0 and world == 1 and not shard and not emul and not args.no_micro:
        try:
            micro = micro_readings(g, cv, pk, torch, np, dev_sync, rnd)
        except Exception as e:                                # noqa: BLE001
            micro = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        stage("stand-alone MSM / NTT readings done")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    parity = "skipped"
    if not args.no_check:
        ok = all(proof == g.prove_closed_form(pk, z, r_, s_) for proof, r_, s_ in results)
        parity = "proof == trapdoor closed form" if ok else "MISMATCH"
        if world > 1:
            f = torch.tensor([0 if ok else 1], device=xdev)
            dist.all_reduce(f)
            if int(f.item()) != 0:
                parity = "MISMATCH"
        assert parity != "MISMATCH", "proof differs from the closed form"
    stage("parity: %s (%d proofs)" % (parity, len(results)))

    if rank == 0:
        N = r1.domain_size
        m, w = r1.m, r1.w
        g1_terms = (N - 1) + (w + 1) + 2 * (m + 4)
        g2_terms = m + 4
        if shard:                      # this rank's launches see 1/world of every query vector
            g1_terms, g2_terms = g1_terms / world, g2_terms / world
        g1b, g2b = 32 + g.sizes["g1"], 32 + g.sizes["g2"]
        alg_bytes_per_proof = g1_terms * g1b + g2_terms * g2b       # dominant kernel only (5 launches)
        alg_bytes_per_launch = alg_bytes_per_proof / 5.0
        avg_launch_ms = acc_ms_sum / max(1, acc_launches)
        achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        prove_alg_bytes = (6 * 64 * N + alg_bytes_per_proof + 32 * m + 8 * int(sum(int(rp[-1]) for rp in r1.row_ptr))
                           + 3 * 32 * n)
        # integer roofline: multiply-adds issued by the accumulation launches / their summed duration
        windows = acc_points / max(1, args.steps) / float(g1_terms + g2_terms)       # table windows per term
        mpa, mpa_source = mads_per_add(args.curve)
        peaks = [v for k, v in mad_peak.items() if not k.endswith("_error") and v and v > 0]
        mad_peak_t = sum(peaks) / len(peaks) if peaks else None
        mads_per_proof = windows * (g1_terms * mpa["g1"] + g2_terms * mpa["g2"])
        acc_ms_per_proof = acc_ms_sum / max(1, args.steps)
        mad_rate_t = mads_per_proof / (acc_ms_per_proof * 1e-3) / 1e12 if acc_ms_per_proof > 0 else 0.0
        traffic, traffic_note = pmc_traffic(n, args.curve)
        mode_txt = ("one proof sharded over %d GPU(s)" % world) if shard else (
            "throughput: %d proofs in flight per GPU, z resident in HBM" % len(ctxs))
        out = {
            "metric": "R1CS constraints/sec (Groth16 prove, %s; %s)" % ("BLS12-381" if args.curve == "bls12_381" else "BN254", mode_txt),
            "value": (1 if shard else world) * n * args.steps / dt,
            "unit": "constraints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic" if not emul else "synthetic -- EMULATOR DRY RUN on CPU (plumbing check, NOT a measurement)",
            "config": {"workload": ("S2 mulchain R1CS, n=%d constraints (N=2^%d), Groth16/%s, " % (n, N.bit_length() - 1, args.curve))
                                   + ("ONE proof per step, MSM term ranges sharded over the GPUs, pk shard+CSR+z resident in HBM"
                                      if shard else "one independent proof stream per GPU, pk+CSR+z resident in HBM"),
                       "parallelism": ("msm-shard x%d (RCCL behind the C ABI: %s)" % (
                                           world, "bucket-level ring reduce-scatter + all-gather" if args.shard_exchange == "ring"
                                           else "all-gather of 5 partial sums"))
                                      if shard else
                                      "replicas x%d (independent proofs, no collective), %d proofs in flight per GPU"
                                      % (world, len(ctxs))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "msm_accumulate_kernel (bucket accumulation, 4 G1 + 1 G2 launches per proof)",
                         "alg_bytes_per_launch": alg_bytes_per_launch, "avg_launch_ms": avg_launch_ms,
                         "single_stream": None if not solo else {
                             "avg_launch_ms": solo, "achieved": alg_bytes_per_launch / (solo * 1e-3) / 1e9,
                             "frac": alg_bytes_per_launch / (solo * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "same kernel, one proof in flight, measured after the timed region"},
                         "note": "integer-ALU bound by construction (~10 Fq mul per 128 B term); see DESIGN.md",
                         "alu": {"unit": "T v_mad_u64_u32/s", "achieved": mad_rate_t, "peak": mad_peak_t or MAD_PEAK_REF_T,
                                 "frac": mad_rate_t / (mad_peak_t or MAD_PEAK_REF_T),
                                 "peak_source": ("measured on this box around the timed region (ark355_diag_mad_rate): %s" % json.dumps(mad_peak))
                                                if mad_peak_t else "reference constant (not measured in this run)",
                                 "windows_per_term": windows,
                                 "mads_per_add": mpa, "mads_per_add_source": mpa_source,
                                 "note": "all 5 accumulation launches of a proof; with several proofs in flight the "
                                         "launches share the chip, so the single-stream run (--inflight 1) is the "
                                         "clean reading"}},
            "parity": parity,
            # box-normalised figures: the same library reads 4-5 % apart on two boxes of the pool (sustained clock under the power
            # cap); these are the headline numbers rescaled to a box that issues MAD_PEAK_REF_T multiply-adds per second, and the
            # time in gfx clock cycles (mean gfxclk of the timed region from the telemetry sampler)
            "box": box_block(gpu_clocks, dt, args.steps, (1 if shard else world) * n * args.steps / dt, n, telemetry),
            # host CPU seconds burnt per second of the timed region by this rank (launch threads, waits, the O(1) proof tail)
            "host_cpu_cores": host_cpu_s / dt if dt > 0 else None,
            "host_cpu_threads": host_cpu_threads,      # cores per thread name over the timed region (top 8)
            # the metric as SURVEY.md 8d defines it (host z -> host proof, single proofs) beside the throughput headline
            "latency": latency,
            # which schedule the library's measured choice settled on for this run, and what it saw while choosing
            "schedule": {"in_timed_region": sched_used, "calibration": calib,
                         "alone": None if shard or pkh is None else g.lib.sched_info(g.ctx, pkh, False)},
            "schedule_ab": ab,
            "isolated": isolated,
            "micro": micro,
            "stream_map": stream_map,
            "telemetry": telemetry,
            # elapsed times of OVERLAPPING stream segments of the last single proof (the pipeline runs them concurrently:
            # they do not add up and are not kernel times -- `isolated` has those)
            "phases_overlapping_stream_segments_ms": tim,
            "prove_alg_bytes": prove_alg_bytes,
            "prove_hbm_frac": prove_alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            # BASELINE's second metric: ISOLATED readings (one proof alone on one stream; `isolated`), not the overlapping
            # stream segments rounds 1-3 derived them from; stand-alone ark355_msm_dev calls are under `micro`
            "msm_g1_mscalar_mul_per_s": isolated["msm_g1_mscalar_mul_per_s"] if isolated else None,
            "msm_g2_mscalar_mul_per_s": isolated["msm_g2_mscalar_mul_per_s"] if isolated else None,
            "ntt_elements_per_s": isolated["ntt_elements_per_s_witness_map"] if isolated else None,
            "prep_s": prep_s,
            # how the resident key sits in HBM (window size, windows, table stride, bytes of its five window tables)
            "key_tables": g.lib.pk_table_info(pkh if pkh is not None else sg.load_pk_shard(pk)),
        }
        # row format of the 28-bit window tables (policy PACK_ROWS): G1 row bytes from the table bytes
        try:
            kt = out["key_tables"]
            rows_g1 = kt["windows"] // max(1, kt["table_stride"]) * (g1_terms + 2.0 * g2_terms)     # a G2 row is two G1-sized halves
            kt["g1_row_bytes"] = round(kt["table_bytes"] / rows_g1) if rows_g1 else None
            kt["pack_rows_policy"] = g.lib.ctx_get_policy(g.ctx, "PACK_ROWS")
            kt["note"] = ("rows: bit-packed (96 B BLS12-381 / 64 B BN254) or one word per limb (128 / 80 B); default: packed when a packed row "
                          "is whole 64-byte sectors or when the key would not fit HBM otherwise (ARK355_PACK_ROWS=1 forces it: tables x0.75 on "
                          "BLS12-381, measured +1.5 % per proof in flight)")
        except Exception:                                     # noqa: BLE001
            pass
        if not args.no_e2e and world == 1 and not emul and not shard:
            stage("e2e (synthesis in the loop) ...")
            out["e2e"] = e2e_reading(args.curve, n, len(ctxs), out["value"], s3=args.e2e_s3)
        if not args.no_cpu_baseline and world == 1 and not emul:
            wl = None
            try:
                if results and not shard:
                    pr, r_, s_ = results[0]
                    sz = g.sizes
                    raw = (pr.a, pr.b, pr.c) if hasattr(pr, "a") else tuple(pr)
                    wl = {"n": n, "ell": r1.ell, "w": r1.w, "mats": [(r1.row_ptr[i], r1.col[i], r1.coeff[i]) for i in range(3)],
                          "z": synthetic.z_to_mont_bytes(cv, z), "r": r_, "s": s_, "proof": raw,
                          "pk": {"a_query": pk.a_query, "b_g1_query": pk.b_g1_query, "b_g2_query": pk.b_g2_query, "h_query": pk.h_query,
                                 "l_query": pk.l_query, "alpha_g1": pk.vk.alpha_g1, "beta_g1": pk.beta_g1, "delta_g1": pk.delta_g1,
                                 "beta_g2": pk.vk.beta_g2, "delta_g2": pk.vk.delta_g2}}
            except Exception as e:                            # noqa: BLE001
                stage("cpu_baseline: the bench's workload could not be handed to the oracle (%s)" % e)
                wl = None
            out["cpu_baseline"] = cpu_baseline(args.curve, workload=wl)
            if out["cpu_baseline"].get("parity_vs_oracle") == "MISMATCH":
                out["parity"] = "MISMATCH against oracle/c"
        if sampler is not None:
            telemetry["host_at_end"] = GT.host_counters()
        if not emul and not args.profile_run:
            try:
                telemetry["dispatch_gap_after"] = g.lib.diag_dispatch(g.ctx)
            except Exception as e:                            # noqa: BLE001
                telemetry["dispatch_gap_after"] = {"error": str(e)[:160]}
        sys.stdout.flush()
        os.write(_JSON_FD if _JSON_FD is not None else 1, (json.dumps(out) + "\n").encode())
    if sampler is not None:
        sampler.stop()
    for c in ctxs[1:]:
        g.lib.ctx_destroy(c)
    if sg is not None:
        sg.close()
    g.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
