# Round profile: one gpurun call.  usage (on the GPU box): bash tools/profile_round.sh r03
# Everything lands in gpurun_out/<round>/; tools/collect_profiles.py turns it into the tracked files under profiles/.
# rocprofv3 gets the program itself after `--` (python3 ...), never a shell or env wrapper (the profiler's preloaded library has
# initialised the GPU before the program starts: any exec hop in between takes the box down).  Counter passes are separate runs with
# --kernel-trace only.
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
G1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES"
G2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
G5="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"

# ---- 1. the default line's kernels: trace of the same command the driver runs (fewer steps, no extras)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
# ... and with the block-ahead oscillators serialised in front of the voices (OW_TREM_SERIAL=1): each kernel's own time
export OW_TREM_SERIAL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -o t -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/trace_serial.log 2>&1
unset OW_TREM_SERIAL

# ---- 2. counters of the default kernels: 4 096-engine pool, one oscillator per engine (the default); the big-pool tremolo and preamp
# kernels forced (a 4 096-engine pool would pick the quad-per-engine variants on its own)
export OW_TREM_WIDE=0
export OW_PREAMP_WIDE=0
for g in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE" "$G5"; do
  n=$(echo $g | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/pmc_$n -o p -- python3 bench.py --steps 8 --warmup 2 --instances 4096 --no-cpu-baseline --no-extras > $O/pmc_$n.log 2>&1
done
unset OW_TREM_WIDE OW_PREAMP_WIDE

# ---- 3. the non-default kernels: kernel trace + instruction-mix / cycle / memory-instruction passes each
prof() {   # prof <tag> <bench args...>
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -o t -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $O/trace_$tag.log 2>&1
}
pmc() {    # pmc <tag> <group name> "<counters>" <bench args...>
  tag=$1; gn=$2; g=$3; shift 3
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/pmc_${tag}_$gn -o p -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $O/pmc_${tag}_$gn.log 2>&1
}
prof melange --preamp melange --instances 65536 --steps 10 --warmup 2
pmc melange insts "$G1" --preamp melange --instances 4096 --steps 6 --warmup 2
pmc melange cycles "$G2" --preamp melange --instances 4096 --steps 6 --warmup 2
pmc melange mem "$G5" --preamp melange --instances 4096 --steps 6 --warmup 2
prof mpa --power-amp melange --instances 65536 --steps 2 --warmup 2 --tremolo-groups 1
pmc mpa insts "$G1" --power-amp melange --instances 2048 --steps 3 --warmup 1 --tremolo-groups 1
pmc mpa cycles "$G2" --power-amp melange --instances 2048 --steps 3 --warmup 1 --tremolo-groups 1
pmc mpa mem "$G5" --power-amp melange --instances 2048 --steps 3 --warmup 1 --tremolo-groups 1
prof batch --workload batch --steps 2 --warmup 1
pmc batch insts "$G1" --workload batch --steps 1 --warmup 0
prof p256 --instances 256 --steps 30 --warmup 3
pmc p256 insts "$G1" --instances 256 --steps 10 --warmup 2
pmc p256 cycles "$G2" --instances 256 --steps 10 --warmup 2
pmc p256 mem "$G5" --instances 256 --steps 10 --warmup 2

# ---- 4. bench lines behind every row of DESIGN section 6
python bench.py > $O/bench_default.log 2>&1
python bench.py --steps 938 --warmup 5 --no-extras > $O/bench_10s.log 2>&1
python bench.py --tremolo-groups 1 --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_shared_phase.log 2>&1
python bench.py --tremolo-groups 4096 --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_groups4096.log 2>&1
python bench.py --instances 65536 --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_65536.log 2>&1
python bench.py --instances 524288 --steps 12 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_524288.log 2>&1
python bench.py --host-rate 96000 --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_config3.log 2>&1
python bench.py --preamp melange --instances 65536 --steps 20 --warmup 3 --no-extras > $O/bench_melange.log 2>&1
OW_MEL_LDS=1 python bench.py --preamp melange --instances 65536 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_melange_lds_matrix.log 2>&1
OW_MEL_RANK1=1 python bench.py --preamp melange --instances 65536 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_melange_rank1.log 2>&1
python bench.py --power-amp melange --steps 4 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_power_amp_melange.log 2>&1
python bench.py --power-amp melange --instances 65536 --steps 4 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_power_amp_melange_65536.log 2>&1
python bench.py --power-amp melange --instances 16384 --steps 4 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_power_amp_melange_16384.log 2>&1
python bench.py --workload batch > $O/bench_batch.log 2>&1
OW_TREM_SERIAL=1 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_trem_serial.log 2>&1
tail -1 $O/bench_default.log | cut -c1-700
ls $O
