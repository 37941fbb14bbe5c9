#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE counter_collection.csv (two separate rocprofv3 --pmc passes) -> profiles/pmc_latest.json.
FETCH_SIZE (KiB) is doubled: on gfx950 it tallies 64 B per 128-B request (MI355X_MICROARCH.md §HBM).  Calibrated by load width with
tools/probes/fetch_calibrate.hip (profiles/r6_c2_fetch_calibrate.log): 4, 8 and 16 bytes per lane and the time-major 8-byte walk all
report exactly half their bytes.
usage: tools/pmc_to_json.py <fetch.csv> <write.csv> <label> > profiles/pmc_latest.json"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("tad::", "")
            acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": sys.argv[3], "units": "bytes per launch; fetch = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction), write = WRITE_SIZE KiB x 1024",
       "kernels": {k: {"fetch_bytes": int(fetch.get(k, 0) * 1024 * 2), "write_bytes": int(write.get(k, 0) * 1024)}
                   for k in sorted(set(fetch) | set(write)) if k.startswith("k_")}}
print(json.dumps(out, indent=1))
