"""Condense a rocprofv3 kernel_stats CSV: short kernel names, totals per step.
usage: python tools/prof_summary.py <stats.csv> [n_steps] [top]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"void (?:at::native::)?([\w:]+)<", name)
    if name.startswith("Cijk_"):
        mt = re.search(r"MT(\d+x\d+x\d+)", name)
        return "hipblaslt_gemm_" + name[5:14] + "_MT" + (mt.group(1) if mt else "")
    if "ndjir::" in name:
        return re.sub(r"\(.*", "", name.replace("void ", ""))[:70]
    if m:
        inner = re.search(r"(\w+_kernel|\w+Functor\w*|\w+functor\w*|CatArray\w+|\w+_impl\w*)", name[len(m.group(0)):])
        return (m.group(1).split("::")[-1] + ":" + (inner.group(1) if inner else ""))[:70]
    return name[:70]


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    agg = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Name"])
            a = agg.setdefault(k, [0, 0.0])
            a[0] += int(row["Calls"])
            a[1] += float(row["TotalDurationNs"])
    total = sum(v[1] for v in agg.values())
    print(f"total kernel time {total / 1e6:.2f} ms over {steps:g} steps = {total / 1e6 / steps:.2f} ms/step")
    print(f"{'kernel':70s} {'calls/step':>10s} {'ms/step':>9s} {'%':>6s} {'avg us':>9s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:70s} {c / steps:10.1f} {t / 1e6 / steps:9.3f} {100 * t / total:6.2f} {t / c / 1e3:9.1f}")


if __name__ == "__main__":
    main()
