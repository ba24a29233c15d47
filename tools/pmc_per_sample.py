#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of tools/profile_round.sh (bench.py --instances 4096, L = 512) to per-wavefront,
per-sample figures for the four render kernels, and write profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

Usage: pmc_per_sample.py <dir with pmc_*/p_counter_collection.csv> <out.md> <hbm_traffic.json>
HBM bytes follow the microarch guide's gfx950 correction: (2 * FETCH_SIZE + WRITE_SIZE) KiB."""
import csv
import glob
import json
import sys
from collections import defaultdict

I, L = 4096, 512


def main():
    src, out_md, out_json = sys.argv[1:4]
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(f"{src}/pmc_*/p_counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("owdev::", "").split("<")[0]
            if name.startswith("k_"):
                acc[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = [("voices", "k_voice_steady", I * 64, L, "host"), ("tremolo", "k_tremolo", I, 2 * L, "96 kHz"),
               ("preamp", "k_preamp", 2 * I, 2 * L, "96 kHz"), ("post", "k_post", 2 * I, L, "host")]
    md = ["| kernel | waves | per wave, per sample of rate | VALU | ADD_F64 | MUL_F64 | FMA_F64 | TRANS_F64 | SALU | SMEM | LDS | VMEM rd | VMEM wr |"
          " wave cycles | VALU-busy cycles | LDS bank conflicts |", "|" + "---|" * 16]
    traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --instances 4096 --steps 8 --warmup 2, "
                         "L=512 (tools/profile_round.sh)",
               "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts 64 B per 128-B request)",
               "instances_measured": I, "kernels": {}}
    for label, name, grid, nsamp, rate in kernels:
        c = {k: sum(v[-6:]) / len(v[-6:]) for k, v in acc[(name, grid)].items()}   # last launches = timed steps
        if not c:
            continue
        w = c["SQ_WAVES"]
        per = lambda x: c.get(x, 0.0) / w / nsamp
        # SQ_WAVE_CYCLES / SQ_ACTIVE_INST_VALU count in units of 4 clocks (one wave64 issue slot)
        md.append(f"| {name} | {w:.0f} | {rate} | {per('SQ_INSTS_VALU'):.1f} | {per('SQ_INSTS_VALU_ADD_F64'):.1f} | "
                  f"{per('SQ_INSTS_VALU_MUL_F64'):.1f} | {per('SQ_INSTS_VALU_FMA_F64'):.1f} | {per('SQ_INSTS_VALU_TRANS_F64'):.1f} | "
                  f"{per('SQ_INSTS_SALU'):.1f} | {per('SQ_INSTS_SMEM'):.2f} | {per('SQ_INSTS_LDS'):.2f} | {per('SQ_INSTS_VMEM_RD'):.2f} | "
                  f"{per('SQ_INSTS_VMEM_WR'):.2f} | {4 * per('SQ_WAVE_CYCLES'):.0f} | {4 * per('SQ_ACTIVE_INST_VALU'):.0f} | "
                  f"{c.get('SQ_LDS_BANK_CONFLICT', 0.0):.3g} |")
        b = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        traffic["kernels"][label] = {"kernel": name, "fetch_kib": c["FETCH_SIZE"], "write_kib": c["WRITE_SIZE"],
                                     "hbm_bytes_per_launch_4096": b, "hbm_bytes_per_engine_launch": b / I}
    open(out_md, "w").write("\n".join(md) + "\n")
    json.dump(traffic, open(out_json, "w"), indent=1)
    print("\n".join(md))


if __name__ == "__main__":
    main()
