#!/usr/bin/env python3
"""Print the A/B logs written by tools/gpu_ab.sh."""
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(os.path.dirname(__file__), '..', 'gpurun_out', 'ab_*.log'))):
    got = False
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print("%-28s %7.2f ms/step  acc_avg=%.3f  %s  %s" % (
                os.path.basename(f), d['ms_per_step'], d['roofline']['avg_launch_ms'],
                {k[:-3]: round(v, 2) for k, v in d['phases_ms'].items() if k != 'h2d_ms'}, d['parity'][:8]))
            got = True
    if not got:
        print(os.path.basename(f), 'NO RESULT:', open(f).read()[-300:])
