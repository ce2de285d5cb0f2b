#!/usr/bin/env python
"""Turn a rocprofv3 results .db (rocpd sqlite) into the kernel-stats summary we commit under profiles/.
usage: python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_bench_kernel_stats.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("| kernel | calls | total_us | avg_us | % |")
print("|---|---:|---:|---:|---:|")
for name, calls, tot, avg, pct in rows:
    short = name.split("(")[0].replace("void ", "")
    print(f"| `{short}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |")
