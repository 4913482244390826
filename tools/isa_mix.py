#!/usr/bin/env python3
"""Instruction mix of selected kernels in a `hipcc --cuda-device-only -S` listing (dev tool)."""
import collections, re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
i = 0
while i < len(lines):
    m = re.match(r'^(_Z\S+):\s', lines[i])
    if m and re.search(pat, m.group(1)):
        name = m.group(1)
        ops = collections.Counter()
        j = i + 1
        while j < len(lines) and not lines[j].startswith('.Lfunc_end'):
            mm = re.match(r'^\s+([a-z][a-z_0-9]+)\b', lines[j])
            if mm:
                ops[mm.group(1)] += 1
            j += 1
        tot = sum(ops.values())
        print(name[:100], 'instructions:', tot)
        for k, v in ops.most_common(18):
            print('    %-28s %6d  %5.1f%%' % (k, v, 100.0 * v / tot))
        i = j
    i += 1
