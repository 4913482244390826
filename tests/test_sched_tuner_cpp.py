"""Host-only unit test of the schedule tuner (snark_amd/csrc/sched_tuner.h; tests/cpp/test_sched_tuner.cpp): the phase scoring of
the in-flight classes, the default-first-and-last rule, the 5 % margin in both classes, stragglers, polluted samples -- replayed
on the readings that misled the earlier scoring rules in round 4 (profiles/r04_run{A,C,D,G}_*)."""
import os
import subprocess

from conftest import ROOT


def test_sched_tuner_rules(tmp_path):
    exe = str(tmp_path / "test_sched_tuner")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "snark_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "test_sched_tuner.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "all checks passed" in out.stdout, (out.stdout, out.stderr)
