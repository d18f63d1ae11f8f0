"""Least-squares fit t(R) = a + b*R per kernel symbol over the rocprofv3 kernel stats that tools/fixed_cost.sh collected
(R = rays per step; per point two eager runs of SA and SB timed steps: per-step numbers are (run B - run A) / (SB - SA), so
one-time launches -- parameter creation, the first packing of the weights -- cancel), beside the fit of the graph-replayed
whole step.   usage: python tools/fixed_cost.py <dir> <SA> <SB> <config> [...]"""
import csv
import glob
import json
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import short  # noqa: E402


def fit(Rs, ts):
    A = np.stack([np.ones(len(Rs)), np.asarray(Rs, float)], 1)
    (a, b), *_ = np.linalg.lstsq(A, np.asarray(ts, float), rcond=None)
    return a, b


def main():
    d, sa, sb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    for conf in sys.argv[4:]:
        files = sorted(glob.glob(os.path.join(d, f"{conf}_R*_s{sb}.csv")), key=lambda p: int(re.search(r"_R(\d+)_s", p).group(1)))
        Rs = [int(re.search(r"_R(\d+)_s", p).group(1)) for p in files]
        per = {}      # symbol -> {R: (calls/step, us/step)}
        for R in Rs:
            for steps, sign in ((sb, 1.0), (sa, -1.0)):
                with open(os.path.join(d, f"{conf}_R{R}_s{steps}.csv")) as f:
                    for row in csv.DictReader(f):
                        k = short(row["Name"])
                        a = per.setdefault(k, {}).setdefault(R, [0.0, 0.0])
                        a[0] += sign * int(row["Calls"]) / (sb - sa)
                        a[1] += sign * float(row["TotalDurationNs"]) / 1e3 / (sb - sa)
        whole = []
        for R in Rs:
            try:
                j = json.loads(open(os.path.join(d, f"{conf}_R{R}.json")).read().strip().splitlines()[-1])
                whole.append(j["ms_per_step"])
            except Exception:
                whole.append(float("nan"))
        print(f"## {conf}: rays per step {Rs}")
        print("whole step, graph replay (ms): " + "  ".join(f"{w:.3f}" for w in whole))
        if not any(np.isnan(whole)):
            a, b = fit(Rs, whole)
            print(f"  t(R) = {a:.3f} ms + {1e3 * b:.2f} us * R     t(4096) / (t(512) + 0.3 ms) = {whole[-1] / (whole[Rs.index(512)] + 0.3):.2f}")
        tot = {R: sum(v.get(R, (0, 0))[1] for v in per.values()) for R in Rs}
        a, b = fit(Rs, [tot[R] for R in Rs])
        print("sum of kernel time (us/step): " + "  ".join(f"{tot[R]:.0f}" for R in Rs) + f"   fit a = {a:.0f} us, b = {b:.2f} us/ray")
        calls = {R: sum(v.get(R, (0, 0))[0] for v in per.values()) for R in Rs}
        print("launches per step: " + "  ".join(f"{calls[R]:.0f}" for R in Rs))
        rows = []
        for k, v in per.items():
            ts = [v.get(R, (0, 0))[1] for R in Rs]
            a, b = fit(Rs, ts)
            rows.append((a, b, k, v))
        print(f"{'kernel symbol':70s} {'n/step@512':>10s} {'a us':>8s} {'b us/ray':>9s} {'us@512':>8s} {'a/t(512)':>8s}")
        for a, b, k, v in sorted(rows, key=lambda r: -r[0])[:45]:
            t512 = v.get(512, (0, 0))
            print(f"{k:70s} {t512[0]:10.1f} {a:8.1f} {b:9.3f} {t512[1]:8.1f} {a / max(t512[1], 1e-9):8.2f}")
        print()


if __name__ == "__main__":
    main()
