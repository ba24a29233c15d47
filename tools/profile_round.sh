# Round profile: one gpurun call.  usage (on the GPU box): bash tools/profile_round.sh r02
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
# the counter passes use a 4 096-engine pool, which would pick the quad-lane tremolo: force the big-pool kernel (one lane per engine).
# One tremolo phase group (the shipping configuration): k_tremolo is a ONE-oscillator launch in these counters.
export OW_TREM_WIDE=0
for g in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $g | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/pmc_$n -o p -- python3 bench.py --steps 8 --warmup 2 --instances 4096 --no-cpu-baseline --no-extras > $O/pmc_$n.log 2>&1
done
unset OW_TREM_WIDE
python bench.py --steps 938 --warmup 5 --no-extras > $O/bench_10s.log 2>&1
python bench.py > $O/bench_default.log 2>&1
python bench.py --instances 65536 --no-extras --no-cpu-baseline > $O/bench_65536.log 2>&1
python bench.py --preamp melange --instances 65536 --steps 20 --warmup 3 --no-extras > $O/bench_melange.log 2>&1
OW_MEL_RANK1=1 python bench.py --preamp melange --instances 65536 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_melange_rank1.log 2>&1
python bench.py --workload batch > $O/bench_batch.log 2>&1
tail -1 $O/bench_default.log | cut -c1-600
ls $O
