#!/usr/bin/env python3
"""Turn what tools/profile_round.sh left under gpurun_out/<round>/ into the tracked files of profiles/.
usage: python tools/collect_profiles.py r03

  profiles/<r>_kernel_stats.csv, <r>_kernel_trace_summary.md      default line (trace/)
  profiles/<r>_kernel_trace_serial.md                             the same with OW_TREM_TRAJ=0: one oscillator per instance (k_tremolo beside the voices), rounds 1-3
  profiles/<r>_pmc_summary.md, <r>_pmc_per_sample.md, hbm_traffic.json   counters of the default kernels (4 096-engine pool)
  profiles/<r>_<tag>_kernel_trace.md, <r>_<tag>_pmc.md            melange preamp / melange power amp / batch / 256-engine pool
  profiles/<r>_bench_<name>.json                                  one file per bench line (the last line of each bench_*.log)
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(pattern):
    got = sorted(glob.glob(pattern, recursive=True))
    return got[0] if got else None


def pmc_table(files):
    acc = defaultdict(lambda: defaultdict(list))
    for path in files:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("owdev::", "")
            if name.startswith("k_"):
                acc[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["| kernel | grid threads | launches | counter | mean per launch |", "|---|---|---|---|---|"]
    derived = ["", "Derived (per wavefront of the launch; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count in units of 4 clocks):", "",
               "| kernel | grid threads | waves | VALU per wave | f64 add / mul / fma / trans per wave | SALU | SMEM | LDS | VMEM rd / wr | wave cycles | VALU-busy cycles | VALU busy |",
               "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for key in sorted(acc, key=lambda k: (k[0], -k[1])):
        c = {k: sum(v) / len(v) for k, v in acc[key].items()}
        for cname, vals in sorted(acc[key].items()):
            lines.append(f"| {key[0]} | {key[1]} | {len(vals)} | {cname} | {sum(vals) / len(vals):.6g} |")
        w = c.get("SQ_WAVES") or max(key[1] // 64, 1)
        g = lambda n: c.get(n, float("nan")) / w
        busy = (c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else float("nan")
        derived.append(f"| {key[0]} | {key[1]} | {w:.0f} | {g('SQ_INSTS_VALU'):.0f} | {g('SQ_INSTS_VALU_ADD_F64'):.0f} / {g('SQ_INSTS_VALU_MUL_F64'):.0f} / "
                       f"{g('SQ_INSTS_VALU_FMA_F64'):.0f} / {g('SQ_INSTS_VALU_TRANS_F64'):.0f} | {g('SQ_INSTS_SALU'):.0f} | {g('SQ_INSTS_SMEM'):.0f} | "
                       f"{g('SQ_INSTS_LDS'):.0f} | {g('SQ_INSTS_VMEM_RD'):.0f} / {g('SQ_INSTS_VMEM_WR'):.0f} | {4 * g('SQ_WAVE_CYCLES'):.0f} | "
                       f"{4 * g('SQ_ACTIVE_INST_VALU'):.0f} | {busy:.2f} |")
    return "\n".join(lines + derived) + "\n"


def main():
    r = sys.argv[1] if len(sys.argv) > 1 else "r04"
    src = os.path.join(ROOT, "gpurun_out", r)
    dst = os.path.join(ROOT, "profiles")
    py = sys.executable
    made = []

    def run(args, out=None):
        res = subprocess.run([py] + args, capture_output=True, text=True)
        if res.returncode != 0:
            print("FAILED", args, res.stderr[-500:])
        if out:
            open(out, "w").write(res.stdout)
            made.append(out)
        return res.stdout

    # default line: trace + stats
    t = find(f"{src}/trace/**/t_kernel_trace.csv")
    if t:
        run([os.path.join(ROOT, "tools", "summarize_profile.py"), t, os.path.join(dst, f"{r}_kernel_trace_summary.md")])
        made.append(os.path.join(dst, f"{r}_kernel_trace_summary.md"))
    s = find(f"{src}/trace/**/t_kernel_stats.csv")
    if s:
        open(os.path.join(dst, f"{r}_kernel_stats.csv"), "w").write(open(s).read())
        made.append(os.path.join(dst, f"{r}_kernel_stats.csv"))
    t = find(f"{src}/trace_serial/**/t_kernel_trace.csv")
    if t:
        run([os.path.join(ROOT, "tools", "summarize_profile.py"), t, os.path.join(dst, f"{r}_kernel_trace_serial.md")])
        made.append(os.path.join(dst, f"{r}_kernel_trace_serial.md"))
    t = find(f"{src}/trace_host/**/t_kernel_trace.csv")
    if t:       # round 5: the same command with --deliver host (k_chain_stream instead of k_preamp + k_post)
        run([os.path.join(ROOT, "tools", "summarize_profile.py"), t, os.path.join(dst, f"{r}_kernel_trace_host_delivery.md")])
        made.append(os.path.join(dst, f"{r}_kernel_trace_host_delivery.md"))
    for name, out in (("restrike_hbm.txt", "restrike_trace.txt"), ("restrike_host.txt", "restrike_trace_host_delivery.txt"), ("smoke.txt", "smoke.txt"),
                      ("pmc_restrike.txt", "pmc_restrike_voice_kernels.txt"), ("probe_release.txt", "probe_release.txt"), ("soak.txt", "soak.txt"), ("soak_gpu.txt", "soak_gpu.txt"),
                      ("probe_soak_variants.txt", "probe_soak_variants.txt"), ("probe_fused.txt", "probe_fused.txt")):
        if os.path.exists(os.path.join(src, name)):
            open(os.path.join(dst, f"{r}_{out}"), "w").write(open(os.path.join(src, name)).read())
            made.append(os.path.join(dst, f"{r}_{out}"))
    # default kernels: counters
    files = [f for f in sorted(glob.glob(f"{src}/pmc_*/**/p_counter_collection.csv", recursive=True))
             if not any(f"pmc_{tag}_" in f for tag in ("melange", "mpa", "batch", "p256"))]
    if files:
        open(os.path.join(dst, f"{r}_pmc_summary.md"), "w").write(pmc_table(files))
        made.append(os.path.join(dst, f"{r}_pmc_summary.md"))
        # per-sample table + hbm_traffic.json want the directory layout pmc_<NAME>/p_counter_collection.csv
        tmp = os.path.join(src, "_flat")
        os.makedirs(tmp, exist_ok=True)
        for f in files:
            name = [p for p in f.split(os.sep) if p.startswith("pmc_")][0]
            os.makedirs(os.path.join(tmp, name), exist_ok=True)
            open(os.path.join(tmp, name, "p_counter_collection.csv"), "w").write(open(f).read())
        run([os.path.join(ROOT, "tools", "pmc_per_sample.py"), tmp, os.path.join(dst, f"{r}_pmc_per_sample.md"), os.path.join(dst, "hbm_traffic.json")])
        made += [os.path.join(dst, f"{r}_pmc_per_sample.md"), os.path.join(dst, "hbm_traffic.json")]
    # non-default kernels
    for tag in ("melange", "mpa", "batch", "p256"):
        t = find(f"{src}/trace_{tag}/**/t_kernel_trace.csv")
        if t:
            run([os.path.join(ROOT, "tools", "summarize_profile.py"), t, os.path.join(dst, f"{r}_{tag}_kernel_trace.md")])
            made.append(os.path.join(dst, f"{r}_{tag}_kernel_trace.md"))
        files = sorted(glob.glob(f"{src}/pmc_{tag}_*/**/p_counter_collection.csv", recursive=True))
        if files:
            open(os.path.join(dst, f"{r}_{tag}_pmc.md"), "w").write(pmc_table(files))
            made.append(os.path.join(dst, f"{r}_{tag}_pmc.md"))
    # bench lines
    for log in sorted(glob.glob(f"{src}/probe_*.log")):
        out = os.path.join(dst, f"{r}_{os.path.basename(log)[:-4]}.txt")
        open(out, "w").write(open(log).read())
        made.append(out)
    for log in sorted(glob.glob(f"{src}/bench_*.log")):
        line = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
        if not line:
            print("no JSON line in", log)
            continue
        name = os.path.basename(log)[len("bench_"):-len(".log")]
        out = os.path.join(dst, f"{r}_bench_{name}.json")
        open(out, "w").write(json.dumps(json.loads(line[-1]), indent=1) + "\n")
        made.append(out)
    print("\n".join(os.path.relpath(m, ROOT) for m in made))
    # a collected file that is a stack trace (or empty) is not evidence: fail, so that the round's README cannot list it as a measurement
    bad = []
    for m in made:
        try:
            txt = open(m, errors="replace").read()
        except OSError:
            bad.append((m, "missing")); continue
        if "Traceback (most recent call last)" in txt:
            bad.append((m, "contains a Python traceback"))
        elif not txt.strip():
            bad.append((m, "empty"))
    if bad:
        for m, why in bad:
            print("NOT EVIDENCE: %s %s" % (os.path.relpath(m, ROOT), why), file=sys.stderr)
        sys.exit(2)


if __name__ == "__main__":
    main()
