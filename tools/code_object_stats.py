#!/usr/bin/env python3
"""Instruction counts of the shipped kernels, read from the BUILT library (no hand-entered constants).

`python tools/code_object_stats.py [libark355.so]` extracts the gfx950 code objects from the library's fat binary
(clang offload bundles), disassembles them with llvm-objdump and reports, for the bucket-accumulation kernels, the
instruction mix of their HOT PATH -- the branch-free runs of at least HOT_MIN instructions, which together are the mixed
addition (XYZZ += affine) of one sorted entry (two runs: up to the test for the exceptional cases P = +-acc, and after it):
every loop iteration executes each of them exactly once; everything else in the kernel (entry fetch, bucket boundary, flush,
the exceptional cases' calls) sits in blocks of a few dozen instructions and is listed as `other`.  snark_amd/build.py writes the result
next to the library (libark355.stats.json); bench.py prices `roofline.alu` with it.

Also reports scratch (spill) accesses per kernel and inside the hot block -- the committed answer to "does the loop spill".
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAD = ("v_mad_u64_u32", "v_mad_i64_i32")
HOT_MIN = 128
# kernels of interest: regex on the mangled name -> key
KERNELS = [
    (r"msm_accumulate28_kernelINS_11BlsFqParams", "bls12_381.g1"),
    (r"msm_accumulate_g2l28_kernelINS_11BlsFqParams", "bls12_381.g2"),
    (r"msm_accumulate28p_kernelINS_11BlsFqParams", "bls12_381.g1_packed"),
    (r"msm_accumulate_g2l28p_kernelINS_11BlsFqParams", "bls12_381.g2_packed"),
    (r"msm_accumulate28_kernelINS_10BnFqParams", "bn254.g1"),
    (r"msm_accumulate_g2l28_kernelINS_10BnFqParams", "bn254.g2"),
    (r"msm_accumulate28p_kernelINS_10BnFqParams", "bn254.g1_packed"),
    (r"msm_accumulate_g2l28p_kernelINS_10BnFqParams", "bn254.g2_packed"),
    (r"ntt_pass_kernelINS_2FpINS_11BlsFrParamsEEELi9E", "bls12_381.ntt_pass9"),
    (r"ntt_pass_kernelINS_2FpINS_11BlsFrParamsEEELi6E", "bls12_381.ntt_pass6"),
    (r"ntt_seam_kernelINS_2FpINS_11BlsFrParamsEEELi6E", "bls12_381.ntt_seam6"),
]


def code_objects(lib_path):
    d = open(lib_path, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", d):
        o = m.start()
        n = struct.unpack_from("<Q", d, o + 24)[0]
        p = o + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, p)
            p += 24
            triple = d[p:p + tl].decode()
            p += tl
            if size and "gfx950" in triple:
                out.append(d[o + off:o + off + size])
    return out


def kernel_listings(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(elf_bytes)
        f.flush()
        txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
    cur, res = None, {}
    for line in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(_Z\S+)>:", line)
        if m:
            cur = m.group(1)
            res[cur] = []
        elif cur is not None and line.startswith("\t"):
            res[cur].append(line.strip().split()[0] if line.strip() else "")
    return res


def kernel_stats(instrs):
    blocks, cur = [], []
    for ins in instrs:
        if not ins:
            continue
        cur.append(ins)
        if ins.startswith("s_cbranch") or ins.startswith("s_branch") or ins.startswith("s_endpgm") or ins.startswith("s_setpc"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    hot = [i for b in blocks if len(b) >= HOT_MIN for i in b]
    def cnt(b, pred):
        return sum(1 for i in b if pred(i))
    allv = [i for b in blocks for i in b]
    return {
        "instructions": len(allv),
        "hot_block": {
            "runs": sorted((len(b) for b in blocks if len(b) >= HOT_MIN), reverse=True),
            "instructions": len(hot),
            "multiply_adds": cnt(hot, lambda i: i in MAD),
            "valu": cnt(hot, lambda i: i.startswith("v_")),
            "scratch_loads": cnt(hot, lambda i: i.startswith("scratch_load")),
            "scratch_stores": cnt(hot, lambda i: i.startswith("scratch_store")),
            "global_loads": cnt(hot, lambda i: i.startswith("global_load") or i.startswith("buffer_load")),
            "lds": cnt(hot, lambda i: i.startswith("ds_")),
        },
        "scratch_loads": cnt(allv, lambda i: i.startswith("scratch_load")),
        "scratch_stores": cnt(allv, lambda i: i.startswith("scratch_store")),
        "multiply_adds": cnt(allv, lambda i: i in MAD),
    }


def library_stats(lib_path):
    found = {}
    for elf in code_objects(lib_path):
        ks = kernel_listings(elf)
        for name, instrs in ks.items():
            for rx, key in KERNELS:
                if key not in found and re.search(rx, name):
                    found[key] = kernel_stats(instrs)
    return found


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "snark_amd", "libark355.so")
    st = library_stats(lib)
    for k in sorted(st):
        s = st[k]
        h = s["hot_block"]
        print("%-24s kernel: %6d instr, %5d mads, scratch ld/st %4d/%4d | hot path %s: %5d instr, %5d mads, %5d VALU, scratch ld/st %3d/%3d, global ld %3d, LDS %3d"
              % (k, s["instructions"], s["multiply_adds"], s["scratch_loads"], s["scratch_stores"], h["runs"], h["instructions"], h["multiply_adds"], h["valu"],
                 h["scratch_loads"], h["scratch_stores"], h["global_loads"], h["lds"]))
    if "--json" in sys.argv:
        print(json.dumps(st))
