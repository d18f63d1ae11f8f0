"""Per-step kernel table from two rocprofv3 kernel-stats CSVs of the same command run with SA and SB timed steps:
(run B - run A) / (SB - SA), so one-time launches (parameter creation, first packing of the weights) cancel exactly.
usage: python tools/diff_summary.py <statsA.csv> <statsB.csv> <SA> <SB> [top]"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import short  # noqa: E402


def main():
    fa, fb, sa, sb = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    top = int(sys.argv[5]) if len(sys.argv) > 5 else 60
    agg = {}
    for path, sign in ((fb, 1.0), (fa, -1.0)):
        with open(path) as f:
            for row in csv.DictReader(f):
                a = agg.setdefault(short(row["Name"]), [0.0, 0.0])
                a[0] += sign * int(row["Calls"]) / (sb - sa)
                a[1] += sign * float(row["TotalDurationNs"]) / (sb - sa)
    agg = {k: v for k, v in agg.items() if v[0] > 1e-9}
    total = sum(v[1] for v in agg.values())
    calls = sum(v[0] for v in agg.values())
    print(f"per step (difference of a {sb}-step and a {sa}-step run): {calls:.0f} launches, {total / 1e6:.3f} ms of kernel time")
    print(f"{'kernel':70s} {'calls/step':>10s} {'ms/step':>9s} {'%':>6s} {'avg us':>9s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:70s} {c:10.1f} {t / 1e6:9.3f} {100 * t / total:6.2f} {t / c / 1e3:9.1f}")


if __name__ == "__main__":
    main()
