#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size) so the bench pool's launches can be read apart
from warm-up / single-instance launches of the same kernels.  Usage: summarize_profile.py <kernel_trace.csv> [out.md]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    rows = defaultdict(list)
    meta = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("owdev::", "")
            key = (name, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))
            rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            meta[key] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    lines = ["| kernel | grid (threads x, y) | calls | avg ms | min ms | max ms | total ms | VGPR | AGPR | SGPR | LDS B | scratch B |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for key, d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        name, gx, gy = key
        m = meta[key]
        lines.append(f"| {name} | {gx} x {gy} | {len(d)} | {sum(d) / len(d):.3f} | {min(d):.3f} | {max(d):.3f} | {sum(d):.1f} | "
                     f"{m[0]} | {m[1]} | {m[2]} | {m[3]} | {m[4]} |")
    # the bench's timed region is the tail of the trace: pool creation / warm-up launches (other block lengths) come first,
    # so the last N launches of each pool-sized kernel are the ones bench.py's HIP events time
    tail = defaultdict(list)
    with open(path) as f:
        rr = sorted(csv.DictReader(f), key=lambda r: int(r["Start_Timestamp"]))
    big = max((int(r["Grid_Size_X"]) for r in rr if "k_voice_steady" in r["Kernel_Name"]), default=0)
    for r in rr:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("owdev::", "")
        if name.startswith("k_") and int(r["Grid_Size_X"]) * 64 >= big:
            tail[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    lines += ["", "Timed region only (last 20 launches of each pool-sized kernel):", "",
              "| kernel | launches | avg ms | min ms | max ms |", "|---|---|---|---|---|"]
    for name, d in sorted(tail.items(), key=lambda kv: -sum(kv[1][-20:])):
        d = d[-20:]
        lines.append(f"| {name} | {len(d)} | {sum(d) / len(d):.3f} | {min(d):.3f} | {max(d):.3f} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
