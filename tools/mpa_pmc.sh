# counters of k_post_mpa on a chip-filling pool: where the cycles go (i-cache, LDS, scratch, scalar)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/mpa_pmc; mkdir -p $O
A="--power-amp melange --instances 16384 --steps 2 --warmup 2 --tremolo-groups 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $O/icache -o p -- python3 bench.py $A > $O/icache.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/active -o p -- python3 bench.py $A > $O/active.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/level -o p -- python3 bench.py $A > $O/level.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/insts -o p -- python3 bench.py $A > $O/insts.log 2>&1
ls $O
