cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01e
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01e/trace -o t -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r01e/trace.log 2>&1
# the counter passes use a 4 096-engine pool, which would pick the quad-lane tremolo: force the big-pool kernel (one lane per engine)
export OW_TREM_WIDE=0
for g in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $g | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d gpurun_out/r01e/pmc_$n -o p -- python3 bench.py --steps 8 --warmup 2 --instances 4096 --no-cpu-baseline > gpurun_out/r01e/pmc_$n.log 2>&1
done
unset OW_TREM_WIDE
python bench.py --steps 938 --warmup 5 > gpurun_out/r01e/bench_10s.log 2>&1
python bench.py > gpurun_out/r01e/bench_default.log 2>&1
tail -1 gpurun_out/r01e/bench_10s.log | cut -c1-400
ls gpurun_out/r01e
