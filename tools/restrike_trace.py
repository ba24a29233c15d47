"""Prints the kernel launches around the re-strike from the rocprofv3 kernel trace tools/restrike_trace.sh left under gpurun_out/restrike."""
import csv
import glob
import os

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "restrike", "trace")
f = glob.glob(os.path.join(root, "**", "t_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "owdev::k_" in r["Kernel_Name"]][-75:]
t0 = int(sel[0]["Start_Timestamp"])
print("# bench.py --steps 10 --warmup 88 --no-extras --no-cpu-baseline under rocprofv3 --kernel-trace (tools/restrike_trace.sh):")
print("# the last 75 launches = the steps around one whole-pool re-strike (131 072 instances x 64 keys); start = ms since the first line")
for r in sel:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("owdev::", "")
    print("%9.3f  %-28s grid %9s  %8.3f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6, nm, r["Grid_Size_X"],
                                               (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
