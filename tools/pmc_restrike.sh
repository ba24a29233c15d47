# Counters of the voice kernels over one whole-pool re-strike (4 096 instances: the same kernels as the big pool, one round of the chip):
# `gpurun -- bash tools/pmc_restrike.sh`.  Prints, per launch of a voice kernel in the ten timed steps, instructions per voice-sample.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_restrike; mkdir -p $O
for g in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  n=$(echo $g | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/$n -o p -- python3 bench.py --steps 10 --warmup 88 --instances 4096 --no-extras --no-cpu-baseline > $O/$n.log 2>&1
done
python3 tools/pmc_restrike.py
