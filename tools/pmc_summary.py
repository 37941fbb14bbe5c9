#!/usr/bin/env python3
"""Per-kernel average of one rocprofv3 --pmc counter (counter_collection.csv) as CSV.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so a wide
coalesced read stream shows HALF its bytes (MI355X_MICROARCH.md §HBM): `corrected_MB` doubles it.
usage: tools/pmc_summary.py <counter_collection.csv> <COUNTER> > profiles/rN_x.csv"""
import collections
import csv
import sys

path, counter = sys.argv[1], sys.argv[2]
acc = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != counter:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc.setdefault(name, []).append(float(r["Counter_Value"]))
fac = 2.0 if counter == "FETCH_SIZE" else 1.0
print("kernel,dispatches,avg_%s_KiB,corrected_MB_per_launch" % counter)
for name, v in acc.items():
    avg = sum(v) / len(v)
    print('"%s",%d,%.1f,%.1f' % (name, len(v), avg, avg * 1024 * fac / 1e6))
