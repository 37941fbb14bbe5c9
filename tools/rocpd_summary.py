#!/usr/bin/env python3
"""Dump the per-kernel summary (calls, total/avg ns, %) of a rocprofv3 rocpd sqlite file as CSV.
usage: tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r1_x_kernel_stats.csv"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
print("kernel,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
    short = name.split("(")[0].replace("void ", "")
    print('"%s",%d,%.3f,%.3f,%.2f' % (short, calls, total, avg, pct))
