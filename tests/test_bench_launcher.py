"""bench.py's multi-rank plumbing on a machine without GPUs: `--gpus 2` outside a launcher must start two ranks itself,
run a collective over them and print ONE JSON line with n_gpus == 2 -- in replica mode (weak scaling, independent
proofs) and in shard mode (strong scaling, RCCL behind the C ABI; here its shared-memory emulation).  --dry-run-emul
swaps libark355.so for the CPU emulator build of the same sources (checker); the numbers are not measurements."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(*extra, **more_env):
    env = dict(os.environ, **more_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-emul", "--steps", "2", "--warmup", "1",
                        *extra], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_gpus_2_spawns_two_replica_ranks():
    d = _run("--gpus", "2")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    assert d["parity"] == "proof == trapdoor closed form"
    assert "EMULATOR DRY RUN" in d["data"] and "replicas x2" in d["config"]["parallelism"]
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "roofline"):
        assert key in d


@pytest.mark.parametrize("exchange", ["ring"])      # the all-gather mode at 2, 3 and 8 ranks: tests/test_distributed_gloo.py
def test_bench_shard_mode_two_ranks(exchange):
    # ARK355_BENCH_EMUL_LATENCY=1 arms the single-proof readings that rank 0 takes alone on a GPU run: in shard mode with
    # several ranks they must be skipped (a sharded proof is a collective; rank 0 proving alone would wait forever)
    d = _run("--gpus", "2", "--mode", "shard", "--shard-exchange", exchange, "--log-n", "7", ARK355_BENCH_EMUL_LATENCY="1")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["latency"] is None
    assert d["parity"] == "proof == trapdoor closed form"
    assert "msm-shard x2" in d["config"]["parallelism"]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-emul", "--gpus", "2"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "--gpus 2 but the launcher started 1 ranks" in r.stderr


@pytest.mark.parametrize("mode", ["replica", "shard"])
def test_bench_gpus_8_ranks(mode):
    """The driver's 8-GPU launch shape on CPU: eight ranks (one per GPU), one JSON line from rank 0, whole-job value.
    Replica: eight independent proof streams, no collective on the data path; shard: ONE proof whose MSM term ranges are
    spread over the eight ranks, exchanged through the (emulated) RCCL behind the C ABI."""
    extra = ["--gpus", "8", "--steps", "1", "--warmup", "1"]
    if mode == "shard":
        extra += ["--mode", "shard", "--log-n", "7"]
    d = _run(*extra)
    assert d["n_gpus"] == 8 and d["steps"] == 1
    assert d["parity"] == "proof == trapdoor closed form"
    assert d["scaling"] == ("strong" if mode == "shard" else "weak")
    assert ("msm-shard x8" if mode == "shard" else "replicas x8") in d["config"]["parallelism"]


def test_bench_line_carries_the_8d_latency_and_host_cpu_fields():
    """The fields round 3 added to the bench line, exercised over the emulator (ARK355_BENCH_EMUL_LATENCY=1 runs the
    host-z latency / in-flight readings that are otherwise GPU-only): SURVEY-8d latency from page-locked and pageable
    host memory, the in-flight throughput from host z, host CPU by thread class, the key's table layout, the named
    source of roofline.traffic, the mode in `metric`."""
    env = dict(os.environ, ARK355_BENCH_EMUL_LATENCY="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-emul", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert "proofs in flight per GPU, z resident in HBM" in d["metric"]
    lat = d["latency"]
    for k in ("host_pinned_z_ms", "host_pageable_z_ms", "device_z_ms", "constraints_per_s_single_proof_host_pinned_z"):
        assert lat[k] > 0
    for k in ("pinned", "pageable"):
        assert lat["inflight_from_host_z"][k]["value"] > 0
    assert set(d["host_cpu_threads"]) == {"proving_threads", "resident_threads", "short_lived_native_threads"}
    assert d["key_tables"]["table_stride"] == 1 and d["key_tables"]["table_bytes"] > 0
    assert "traffic_source" in d["roofline"] and d["roofline"]["bound"] == "hbm"
    assert "[bench" in r.stderr and "parity" in r.stderr            # stage markers go to stderr, one JSON line to stdout
