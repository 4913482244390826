#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (dev tool)."""
import re
import subprocess
import sys

for f in sys.argv[1:]:
    txt = open(f).read()
    blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
    for b in blocks:
        name = b.split(' ')[0]

        def g(k):
            m = re.search(k + r': (\d+)', b)
            return m.group(1) if m else '?'
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r'ark355::', '', dn)
        dn = re.sub(r'\(.*', '', dn)[:100]
        print("%-100s VGPR=%4s AGPR=%3s SGPR=%3s scratch=%5s occ=%s lds=%s" % (
            dn, g('VGPRs'), g('AGPRs'), g('TotalSGPRs'), g(r'ScratchSize \[bytes/lane\]'),
            g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
