"""Joins a rocprofv3 kernel trace with its HIP runtime API trace (same clock) for ONE steady-state Newton step: for every launch,
WHEN the host issued it against when the previous kernel ended — a launch issued after the previous kernel's end means the GPU
waited for the host. usage: python tools/step_host_timeline.py <…_kernel_trace.csv> <…_hip_api_trace.csv>"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"^void ", "", name)[:60]


def main():
    kern = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Correlation_Id")))
    api = {}
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            if "Launch" in r["Function"]:
                api[r.get("Correlation_Id")] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"])
    kern.sort()
    starts = [i for i, r in enumerate(kern) if r[2].startswith(("k_bratu_jac", "k_bratu_residual_jac"))]
    if len(starts) < 4:
        print("no steps found")
        return
    a, b = starts[-3], starts[-2]
    step = kern[a:b + 2]
    t0 = step[0][0]
    print("# one steady-state Newton step under rocprofv3 --kernel-trace --hip-runtime-trace: host launch call vs kernel execution (µs from the step's first kernel)\n")
    print("| # | kernel | host launch call (start – end) | previous kernel ended | kernel start | idle before | host late by |")
    print("|---:|---|---:|---:|---:|---:|---:|")
    prev_end = None
    for i, (s, e, n, cid) in enumerate(step):
        h = api.get(cid)
        hs = f"{(h[0] - t0) / 1e3:.1f} – {(h[1] - t0) / 1e3:.1f}" if h else "?"
        idle = 0.0 if prev_end is None else (s - prev_end) / 1e3
        late = "" if (h is None or prev_end is None) else f"{max(0.0, (h[1] - prev_end) / 1e3):.1f}"
        pe = "" if prev_end is None else f"{(prev_end - t0) / 1e3:.1f}"
        print(f"| {i} | `{n}` | {hs} | {pe} | {(s - t0) / 1e3:.1f} | {idle:.1f} | {late} |")
        prev_end = e


if __name__ == "__main__":
    main()
