"""Reads a rocprofv3 kernel-trace CSV and prints the timeline of one steady-state Newton step (see step_timeline.sh)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:60]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # a Newton step starts with the Jacobian fill (k_bratu_jac); take the last but one complete step
    starts = [i for i, r in enumerate(rows) if r[2].startswith(("k_bratu_jac", "k_bratu_residual_jac"))]
    if len(starts) < 4:
        print("no steps found")
        return
    a, b = starts[-3], starts[-2]
    step = rows[a:b]
    t0 = step[0][0]
    total = rows[b][0] - t0
    busy = sum(e - s for s, e, _ in step)
    print(f"# one fixed-work Newton step: {len(step)} launches, {total / 1e3:.1f} µs from the first kernel's start to the next step's, "
          f"{busy / 1e3:.1f} µs inside kernels, {(total - busy) / 1e3:.1f} µs between them\n")
    print("| # | kernel | start µs | duration µs | gap before µs |")
    print("|---:|---|---:|---:|---:|")
    prev_end = None
    gaps = {}
    for i, (s, e, n) in enumerate(step):
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        gaps.setdefault(n, []).append(gap)
        print(f"| {i} | `{n}` | {(s - t0) / 1e3:.2f} | {(e - s) / 1e3:.2f} | {gap:.2f} |")
        prev_end = e
    print(f"| | (next step's first kernel) | {total / 1e3:.2f} | | {(rows[b][0] - prev_end) / 1e3:.2f} |")
    print("\n## gap in front of each kernel family (sum / count)\n")
    for n, g in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        print(f"* `{n}`: {sum(g):.1f} µs over {len(g)} launches")


if __name__ == "__main__":
    main()
