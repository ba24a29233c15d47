"""Reads gpurun_out/pmc_restrike/*/p_counter_collection.csv (tools/pmc_restrike.sh): per launch of a voice kernel, the counters and the
kernel's duration, in dispatch order from the re-strike on."""
import csv, glob, collections, sys
rows = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmc_restrike/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_voice" not in name and "k_apply_ops" not in name:
            continue
        key = (int(r["Dispatch_Id"]), name.split("(")[0].replace("owdev::", "").replace("void ", ""), int(r.get("Grid_Size", 0) or 0))
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
keys = sorted(rows)
# the re-strike: from the first k_apply_ops with a big grid on
start = next((i for i, k in enumerate(keys) if "k_apply_ops" in k[1] and k[2] > 64 * 64), 0)
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY"]
print("%-8s %-30s %9s " % ("dispatch", "kernel", "grid") + " ".join("%12s" % n[3:][:12] for n in names))
for k in keys[max(start - 2, 0):start + 16]:
    c = rows[k]
    print("%-8d %-30s %9d " % k + " ".join("%12.4g" % c.get(n, float("nan")) for n in names))
