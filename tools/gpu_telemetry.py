#!/usr/bin/env python3
"""Clock / power / host telemetry for bench.py ("make the driver line reproduce"): enough to tell a slow BOX
from a slow TREE from the bench line alone.

  snapshot(bus_id)       static facts: ROCm / driver versions, compute + memory partition, performance level, power cap,
                         clock ranges, the HIP runtime knobs in the environment, host CPU quota
  Sampler(bus_id, ms)    a child process that reads amdsmi's gpu_metrics every `ms` milliseconds (sclk per XCD, mclk,
                         socket power, hotspot / memory temperature, throttle status, activity) and appends JSON lines;
                         window(t0, t1) summarises the samples of one wall-clock window
  host_counters()        cgroup CPU throttling counters and load average (read before and after a timed region)

Everything is best effort: a missing library, a missing sysfs file or a denied ioctl gives {"error": ...}, never an
exception -- the bench must not depend on it.  Dev / measurement tool; nothing under snark_amd/ imports it.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile
import time


def _smi():
    import amdsmi
    amdsmi.amdsmi_init()
    return amdsmi


def _handle(amdsmi, bus_id):
    hs = amdsmi.amdsmi_get_processor_handles()
    if not hs:
        raise RuntimeError("no amdsmi processors")
    if bus_id:
        want = str(bus_id).lower()
        for h in hs:
            try:
                if str(amdsmi.amdsmi_get_gpu_device_bdf(h)).lower().endswith(want[-10:]):
                    return h
            except Exception:
                pass
    return hs[0]


def _try(fn, *a):
    try:
        v = fn(*a)
        return v
    except Exception as e:           # noqa: BLE001 - telemetry is best effort
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}


def _plain(v):
    """amdsmi returns enums / nested dicts / 'N/A' strings: make it JSON"""
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, (int, float, str, bool)) or v is None:
        return v
    return str(v)


def host_counters():
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                if k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time", "usage_usec"):
                    out[k] = int(v)
            break
        except OSError:
            continue
    try:
        out["loadavg"] = [float(x) for x in open("/proc/loadavg").read().split()[:3]]
    except OSError:
        pass
    try:
        mhz = [float(line.split(":")[1]) for line in open("/proc/cpuinfo") if line.startswith("cpu MHz")]
        if mhz:
            out["cpu_mhz"] = {"min": min(mhz), "mean": sum(mhz) / len(mhz), "max": max(mhz), "n": len(mhz)}
    except (OSError, ValueError):
        pass
    return out


def cpu_quota():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q < 0 else q / p
    except (OSError, ValueError):
        return None


def snapshot(bus_id=None):
    out = {"env": {k: os.environ[k] for k in sorted(os.environ) if k.startswith(("GPU_", "HIP_", "HSA_", "ROCR_", "AMD_", "ARK355_"))},
           "host": {"hw_threads": os.cpu_count(), "cpu_quota_cores": cpu_quota()}}
    try:
        out["rocm_version_file"] = open("/opt/rocm/.info/version").read().strip()
    except OSError:
        pass
    try:
        out["kfd_driver"] = open("/sys/module/amdgpu/version").read().strip()
    except OSError:
        pass
    try:
        smi = _smi()
        h = _handle(smi, bus_id)
    except Exception as e:           # noqa: BLE001
        out["amdsmi"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
        return out
    g = {}
    g["bdf"] = _plain(_try(smi.amdsmi_get_gpu_device_bdf, h))
    g["driver"] = _plain(_try(smi.amdsmi_get_gpu_driver_info, h))
    g["compute_partition"] = _plain(_try(smi.amdsmi_get_gpu_compute_partition, h))
    g["memory_partition"] = _plain(_try(smi.amdsmi_get_gpu_memory_partition, h))
    g["perf_level"] = _plain(_try(smi.amdsmi_get_gpu_perf_level, h))
    g["power_cap"] = _plain(_try(smi.amdsmi_get_power_cap_info, h))
    try:
        g["clock_gfx"] = _plain(smi.amdsmi_get_clock_info(h, smi.AmdSmiClkType.GFX))
        g["clock_mem"] = _plain(smi.amdsmi_get_clock_info(h, smi.AmdSmiClkType.MEM))
    except Exception as e:           # noqa: BLE001
        g["clock_info"] = {"error": str(e)[:120]}
    g["violations"] = _plain(_try(smi.amdsmi_get_violation_status, h))
    out["amdsmi"] = g
    return out


_KEYS = ("current_gfxclk", "current_gfxclks", "current_uclk", "current_socclk", "current_socket_power", "average_socket_power",
         "temperature_hotspot", "temperature_mem", "throttle_status", "indep_throttle_status", "average_gfx_activity",
         "average_umc_activity", "gfx_activity_acc", "accumulation_counter", "prochot_residency_acc", "ppt_residency_acc",
         "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc")


def _sample_loop(bus_id, period_s, path):
    smi = _smi()
    h = _handle(smi, bus_id)
    with open(path, "a", buffering=1) as f:
        while True:
            t = time.time()
            try:
                m = smi.amdsmi_get_gpu_metrics_info(h)
                rec = {"t": t}
                for k in _KEYS:
                    if k in m:
                        rec[k] = _plain(m[k])
                f.write(json.dumps(rec) + "\n")
            except Exception as e:           # noqa: BLE001
                f.write(json.dumps({"t": t, "error": str(e)[:120]}) + "\n")
                time.sleep(0.5)
            d = period_s - (time.time() - t)
            if d > 0:
                time.sleep(d)


class Sampler:
    def __init__(self, bus_id=None, period_ms=25):
        self.path = tempfile.mktemp(prefix="ark355_telemetry_", suffix=".jsonl")
        self.proc = None
        try:
            self.proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--loop", str(bus_id or ""),
                                          str(period_ms / 1e3), self.path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:            # noqa: BLE001
            self.proc = None

    def samples(self):
        out = []
        try:
            for line in open(self.path):
                try:
                    out.append(json.loads(line))
                except ValueError:
                    pass
        except OSError:
            pass
        return out

    def window(self, t0, t1):
        """min / mean / max of every numeric field over the samples with t0 <= t <= t1 (wall clock, time.time())"""
        rows = [r for r in self.samples() if t0 <= r.get("t", 0) <= t1 and "error" not in r]
        out = {"samples": len(rows)}
        if not rows:
            errs = [r["error"] for r in self.samples() if "error" in r]
            if errs:
                out["error"] = errs[-1]
            return out

        def nums(v):
            if isinstance(v, bool):
                return []
            if isinstance(v, (int, float)):
                return [v]
            if isinstance(v, list):
                return [x for x in v if isinstance(x, (int, float)) and not isinstance(x, bool) and x not in (65535, 4294967295)]
            return []
        for k in _KEYS:
            vals = [x for r in rows for x in nums(r.get(k))]
            if not vals:
                continue
            if k.endswith("_acc") or k == "accumulation_counter":
                out[k] = {"first": vals[0], "last": vals[-1]}
            else:
                out[k] = {"min": min(vals), "mean": round(sum(vals) / len(vals), 1), "max": max(vals)}
        return out

    def stop(self):
        if self.proc is not None:
            self.proc.terminate()
            try:
                self.proc.wait(timeout=2)
            except Exception:        # noqa: BLE001
                self.proc.kill()
            self.proc = None
        try:
            os.unlink(self.path)
        except OSError:
            pass


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "--loop":
        try:
            _sample_loop(sys.argv[2] or None, float(sys.argv[3]), sys.argv[4])
        except KeyboardInterrupt:
            pass
        except Exception as e:       # noqa: BLE001
            with open(sys.argv[4], "a") as f:
                f.write(json.dumps({"t": time.time(), "error": "%s: %s" % (type(e).__name__, str(e)[:160])}) + "\n")
    else:
        print(json.dumps(snapshot(), indent=1))
        s = Sampler(period_ms=50)
        time.sleep(1.0)
        print(json.dumps(s.window(0, time.time() + 1), indent=1))
        s.stop()
