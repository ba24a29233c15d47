# Counters of the steady voice kernel on voices that share one 16-sample jitter grid (the plain loop) and on voices spread over all sixteen
# (skewed lane clocks): `gpurun -- bash tools/skew_pmc.sh`, result in gpurun_out/skew/summary.md
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/skew; rm -rf $O; mkdir -p $O
for g in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU_TRANS_F64" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAVES"; do
n=$(echo $g | cut -d' ' -f1)
timeout 600 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/pmc_$n -o p -- python3 tools/probe_jitter_phase.py 4096 512 > $O/log_$n.txt 2>&1
done
python3 - <<'PY' > gpurun_out/skew/summary.md
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for f in glob.glob("gpurun_out/skew/pmc_*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("owdev::","")
        if "k_voice_steady" not in k or int(r["Grid_Size"]) != 4096*64: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel (4 096 engines x 64 voices, 512-sample blocks) | launches | per wavefront and sample: VALU | transcendental f64 | SALU | LDS | wave cycles | VALU-active cycles | waiting (s_waitcnt) | LDS bank-conflict cycles |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k,c in sorted(acc.items()):
    w=sum(c["SQ_WAVES"])/len(c["SQ_WAVES"]); n=len(c["SQ_INSTS_VALU"])
    g=lambda x,m=1: m*sum(c[x])/len(c[x])/w/512 if c.get(x) else float("nan")
    print("| %s | %d | %.1f | %.2f | %.1f | %.2f | %.0f | %.0f | %.0f | %.1f |"%(k,n,g("SQ_INSTS_VALU"),g("SQ_INSTS_VALU_TRANS_F64"),g("SQ_INSTS_SALU"),g("SQ_INSTS_LDS"),g("SQ_WAVE_CYCLES",4),g("SQ_ACTIVE_INST_VALU",4),g("SQ_WAIT_ANY",4),g("SQ_LDS_BANK_CONFLICT")))
PY
cat gpurun_out/skew/summary.md; grep "instances" gpurun_out/skew/log_SQ_INSTS_VALU.txt
