#!/usr/bin/env python3
"""Analyse a rocprofv3 --kernel-trace CSV of bench.py: GPU busy fraction, accumulate-kernel concurrency and
per-kernel sums over the timed proofs (dev tool).  usage: trace_analyze.py trace.csv n_proofs"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nproofs = int(sys.argv[2])
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']),
       re.sub(r'\(.*', '', r['Kernel_Name']).replace('ark355::', '').replace('void ', '')[:46]) for r in rows]
ev.sort()
acc = [e for e in ev if 'accumulate' in e[2]]
last = acc[-5 * nproofs:]
t0, t1 = min(e[0] for e in last), max(e[1] for e in last)
sel = [e for e in ev if e[1] > t0 and e[0] < t1]


def union(f):
    b = 0
    cs = ce = None
    for s, e, n in sel:
        if not f(n):
            continue
        s, e = max(s, t0), min(e, t1)
        if ce is None or s > ce:
            if ce is not None:
                b += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if ce:
        b += ce - cs
    return b / 1e6


span = (t1 - t0) / 1e6
print("window %.1f ms = %.2f ms/proof; any kernel running %.1f%%; an accumulate kernel running %.1f%%" % (
    span, span / nproofs, 100 * union(lambda n: True) / span, 100 * union(lambda n: 'accumulate' in n) / span))
per = collections.defaultdict(float)
cnt = collections.Counter()
for s, e, n in sel:
    per[n] += (min(e, t1) - max(s, t0)) / 1e6
    cnt[n] += 1
for n, v in sorted(per.items(), key=lambda x: -x[1])[:18]:
    print("  %-48s per-proof sum=%7.2f ms  calls/proof=%5.1f avg=%.3f" % (n, v / nproofs, cnt[n] / nproofs, v / cnt[n]))
pts = []
for s, e, n in sel:
    if 'accumulate' in n:
        pts.append((max(s, t0), 1))
        pts.append((min(e, t1), -1))
pts.sort()
c = 0
lastt = pts[0][0]
hist = collections.defaultdict(float)
for t, d in pts:
    hist[c] += (t - lastt) / 1e6
    lastt = t
    c += d
print("accumulate-kernel concurrency, ms per proof:", {k: round(v / nproofs, 2) for k, v in sorted(hist.items())})
