#!/usr/bin/env python3
"""tools/gpu_placement_r5.sh: per candidate allocation, the times the probe printed under the profiler next to the counters of its k_pattern<0>
(partition-major) dispatches, and the correlation of every counter with the time over the candidates.
usage: python tools/placement_counters.py gpurun_out/<tag>"""
import csv
import glob
import os
import sys
from collections import defaultdict

O = sys.argv[1]
for d in sorted(glob.glob(os.path.join(O, "p[0-9]*"))):
    i = os.path.basename(d)[1:]
    log = os.path.join(O, "probe_pmc%s.log" % i)
    rows = [l.split() for l in open(log) if l[:1].isdigit() and "0x" in l]
    pm = [float(r[2]) for r in rows]
    wm = [float(r[3]) for r in rows]
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f or not pm:
        print("set", i, ": no counter file / no probe output")
        continue
    per = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values in dispatch order
    order = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        per[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for mode, times in (("k_pattern<0>", pm), ("k_pattern<1>", wm)):
        ks = [k for k in per if k.replace(" ", "").startswith("voidk_pattern<%s>" % mode[-2]) or k.replace(" ", "").startswith("k_pattern<%s>" % mode[-2])]
        if not ks:
            continue
        k = ks[0]
        print("set %s, %s (%s), times per candidate: %s" % (i, mode, "partition-major" if mode[-2] == "0" else "workgroup-major", " ".join("%.4f" % t for t in times)))
        for c, vals in sorted(per[k].items()):
            vals.sort()
            v = [x for _, x in vals]
            nc = len(times)
            per_c = len(v) // nc if nc else 0
            if per_c == 0:
                continue
            last = [v[j * per_c + per_c - 1] for j in range(nc)]      # the timed (last) dispatch of each candidate
            mt, mv = sum(times) / nc, sum(last) / nc
            cov = sum((a - mt) * (b - mv) for a, b in zip(times, last))
            var = (sum((a - mt) ** 2 for a in times) * sum((b - mv) ** 2 for b in last)) ** 0.5
            print("   %-50s corr %+.2f  %s" % (c, cov / var if var else 0.0, " ".join("%.3g" % x for x in last)))
