#!/usr/bin/env python3
"""Issue-rate model of the bucket-accumulation kernels: cycles per mixed addition from the ISA listing of the
hot loop (wave-wide v_mad_u64_u32 = 5.5 cycles, the chip's measured 28.8 T multiply-adds/s; every other VALU instruction
= 4 cycles, a wave64 on a 16-lane SIMD), times the additions of a proof, against the measured kernel times of a rocprofv3
--kernel-trace --stats CSV.  Dev tool: needs hipcc (the listing is made here, no GPU) and a profiles/*kernel_stats.csv.

usage: tools/issue_model.py [profiles/r02_final_serial_kernel_stats.csv] [proofs_in_the_profile=7]"""
import csv
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CSV = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_final_serial_kernel_stats.csv")
PROOFS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
MAD_CYCLES, VALU_CYCLES = 5.5, 4.0
SIMDS, CLOCK = 256 * 4, 2.4e9
N = 1 << 20                       # S2 at 2^20: m = n + 2 variables, domain 2^21
WINDOWS = 16


def hot_blocks(listing, kernel_re):
    """(instructions, multiply-adds) of the basic blocks that hold the mixed addition: every block with >= 500
    multiply-adds."""
    lines = open(listing).read().split("\n")
    pat = re.compile(kernel_re)
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l) and pat.search(l)][0]
    end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
    blocks, cur = [], [0, 0]
    for l in lines[start:end]:
        if re.match(r"^\.LBB\d+_\d+:", l) or re.match(r"^\s+s_cbranch|^\s+s_branch", l):
            blocks.append(cur)
            cur = [0, 0]
        if re.match(r"^\s+[a-z]", l):
            cur[0] += 1
            cur[1] += "v_mad_u64_u32" in l
    blocks.append(cur)
    hot = [b for b in blocks if b[1] >= 500]
    return sum(b[0] for b in hot), sum(b[1] for b in hot)


def main():
    with tempfile.TemporaryDirectory() as tmp:
        lst = os.path.join(tmp, "bls.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-DNDEBUG", "-Wno-unused-result",
                               "-I", os.path.join(ROOT, "snark_amd", "csrc"), "--cuda-device-only", "-S",
                               os.path.join(ROOT, "snark_amd", "csrc", "ark355_bls.hip"), "-o", lst],
                              stderr=subprocess.DEVNULL)
        g1 = hot_blocks(lst, "msm_accumulate28_kernel")
        g2 = hot_blocks(lst, "msm_accumulate_g2l28_kernel")
    rows = {r["Name"]: r for r in csv.DictReader(open(CSV))}

    def measured(sub):
        r = [v for k, v in rows.items() if sub in k][0]
        return float(r["TotalDurationNs"]) / 1e6 / PROOFS

    m = N + 2
    adds_g1 = ((2 * N - 1) + 3 * (m + 4)) * WINDOWS          # H + A + B1 + L' terms, one addition per (term, window)
    adds_g2 = (m + 4) * WINDOWS
    for name, (ins, mads), adds, lanes, sub in (("G1 msm_accumulate28_kernel", g1, adds_g1, 1, "msm_accumulate28_kernel"),
                                                ("G2 msm_accumulate_g2l28_kernel", g2, adds_g2, 2, "msm_accumulate_g2l28_kernel")):
        cyc = mads * MAD_CYCLES + (ins - mads) * VALU_CYCLES            # per wave and addition (per lane of a pair for G2)
        model_ms = adds * lanes / 64 * cyc / (SIMDS * CLOCK) * 1e3
        meas = measured(sub)
        print("%-32s hot loop: %5d instructions, %5d multiply-adds per %s -> %6.0f cycles per wave; %5.1f M additions per proof: "
              "model %6.2f ms, measured %6.2f ms (%.0f %% of the model's rate); multiply-adds alone at 28.8 T/s: %5.2f ms"
              % (name, ins, mads, "addition" if lanes == 1 else "lane and addition", cyc, adds / 1e6, model_ms, meas,
                 100 * model_ms / meas, adds * lanes * mads / 28.8e12 * 1e3))


if __name__ == "__main__":
    main()
