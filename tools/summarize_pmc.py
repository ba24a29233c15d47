#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per launch for the bench pool's kernels.
Usage: summarize_pmc.py <grid_threads_of_one_engine_wave_kernel> <csv>...   (writes markdown to stdout)"""
import csv
import sys
from collections import defaultdict


def main():
    files = sys.argv[1:]
    acc = defaultdict(lambda: defaultdict(list))
    for path in files:
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("owdev::", "")
                if not name.startswith("k_"):
                    continue
                key = (name, int(r["Grid_Size"]))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("| kernel | grid threads | launches | counter | mean per launch |")
    print("|---|---|---|---|---|")
    for key in sorted(acc, key=lambda k: (k[0], -k[1])):
        for cname, vals in sorted(acc[key].items()):
            print(f"| {key[0]} | {key[1]} | {len(vals)} | {cname} | {sum(vals) / len(vals):.6g} |")


if __name__ == "__main__":
    main()
