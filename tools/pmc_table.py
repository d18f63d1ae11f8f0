"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per launch.
usage: python tools/pmc_table.py <counter_collection.csv> [kernel-substring]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else "ndjir"
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if sub not in k:
                continue
            k = k.split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
    for k, c in acc.items():
        n = len(disp[k])
        print(f"{k}  launches={n}")
        for name, v in sorted(c.items()):
            print(f"    {name:32s} {v / n:16.1f}")


if __name__ == "__main__":
    main()
