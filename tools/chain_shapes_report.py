"""Summarise a rocprofv3 kernel-trace CSV of tools/chain_shapes.py: per net shape, the chain / wgrad kernel durations
of the last repetition.  usage: python tools/chain_shapes_report.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["geometric", "base_colour", "photogrammetric", "roughness", "env_light", "soft_vis", "background"]
# a repetition = one forward chain ... ; split the stream at forward-chain launches (mode 0)
groups, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if re.search(r"k_chain[\dw]?<0,|k_chainp(_nets)?<0>|k_mlp_chain<0,", n):
        cur = {"fwd": d, "bwd": 0.0, "wgrad": 0.0, "other": 0.0}
        groups.append(cur)
    elif cur is not None:
        if re.search(r"k_chain[\dw]?<1,|k_chainp(_nets)?<1>|k_mlp_chain<1,", n): cur["bwd"] += d
        elif "k_wgrad" in n: cur["wgrad"] += d
        elif "ndjir" in n: cur["other"] += d
reps = len(groups) // len(names)
for i, nm in enumerate(names):
    g = groups[i * reps + reps - 1]
    print(f"{nm:16s} fwd {g['fwd']:7.1f} us  bwd {g['bwd']:7.1f} us  wgrad {g['wgrad']:7.1f} us  other ndjir {g['other']:6.1f} us")
