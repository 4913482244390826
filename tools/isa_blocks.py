#!/usr/bin/env python3
"""Per-basic-block counts of multiply-adds, scratch accesses and calls in one kernel of a device listing (dev tool).
usage: isa_blocks.py listing.s kernel_name_regex"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pat = re.compile(sys.argv[2])
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S+:', l) and pat.search(l)][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
blk = 'entry'
stats, order = {}, []
n_split = [0]
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blk = m.group(1)
    if blk not in stats:
        stats[blk] = [0, 0, 0, 0, 0]
        order.append(blk)
    st = stats[blk]
    if re.match(r'^\s+[a-z]', l):
        st[4] += 1
    st[0] += 'v_mad_u64_u32' in l
    st[1] += 'scratch_store' in l
    st[2] += 'scratch_load' in l
    st[3] += 's_swappc' in l
    if re.match(r'^\s+s_cbranch|^\s+s_branch', l):          # fall-through code after a branch is a new block
        n_split[0] += 1
        blk = blk.split('+')[0] + '+%d' % n_split[0]
for b in order:
    s = stats[b]
    if s[0] or s[1] or s[2] or s[3]:
        print("%-12s instr=%5d mads=%5d scratch_st=%3d scratch_ld=%3d calls=%d" % (b, s[4], s[0], s[1], s[2], s[3]))
