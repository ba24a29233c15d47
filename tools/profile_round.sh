# Round profile: one gpurun call.  usage (on the GPU box): bash tools/profile_round.sh r04
# Everything lands in gpurun_out/<round>/; tools/collect_profiles.py turns it into the tracked files under profiles/.
# rocprofv3 gets the program itself after `--` (python3 ...), never a shell or env wrapper (the profiler's preloaded library has
# initialised the GPU before the program starts: any exec hop in between takes the box down).  Counter passes are separate runs with
# --kernel-trace only.
R=${1:-r06}
PART=${2:-ABC}      # A = traces + default counters, B = non-default kernels, C = bench lines (one gpurun call each: a hung command then costs one part)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
G1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES"
G2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
G5="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"

if [[ $PART == *A* ]]; then
# ---- 1. the default line's kernels: trace of the same command the driver runs (fewer steps, no extras).  Since round 4 every instance
# reads the shared tremolo trajectory: no pool-sized tremolo kernel, k_voice_steady's interval is its own
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
# ... and the per-instance oscillators of rounds 1-3 (OW_TREM_TRAJ=0 at pool creation: k_tremolo beside the voices)
export OW_TREM_TRAJ=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -o t -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/trace_serial.log 2>&1
unset OW_TREM_TRAJ
# ... and with every block delivered into a pinned host block (k_chain_stream instead of k_preamp + k_post, round 5)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_host -o t -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --deliver host > $O/trace_host.log 2>&1

# ---- 2. counters of the default kernels: 4 096-engine pool; the big-pool preamp / output kernels forced (a 4 096-engine pool would pick
# the quad-per-engine preamp and the fused chain on its own)
export OW_TREM_WIDE=0
export OW_PREAMP_WIDE=0
export OW_CHAIN_FUSED=0
for g in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE" "$G5"; do
  n=$(echo $g | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/pmc_$n -o p -- python3 bench.py --steps 8 --warmup 2 --instances 4096 --no-cpu-baseline --no-extras > $O/pmc_$n.log 2>&1
done
unset OW_TREM_WIDE OW_PREAMP_WIDE OW_CHAIN_FUSED

fi
if [[ $PART == *B* ]]; then
# ---- 3. the non-default kernels: kernel trace + instruction-mix / cycle / memory-instruction passes each
prof() {   # prof <tag> <bench args...>
  tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -o t -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $O/trace_$tag.log 2>&1
}
pmc() {    # pmc <tag> <group name> "<counters>" <bench args...>
  tag=$1; gn=$2; g=$3; shift 3
  timeout 900 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/pmc_${tag}_$gn -o p -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $O/pmc_${tag}_$gn.log 2>&1
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
pmc batch cycles "$G2" --workload batch --steps 1 --warmup 0
pmc batch mem "$G5" --workload batch --steps 1 --warmup 0
prof p256 --instances 256 --steps 30 --warmup 3
pmc p256 insts "$G1" --instances 256 --steps 10 --warmup 2
pmc p256 cycles "$G2" --instances 256 --steps 10 --warmup 2
pmc p256 mem "$G5" --instances 256 --steps 10 --warmup 2

fi
if [[ $PART == *C* ]]; then
# ---- 4. bench lines behind every row of DESIGN section 6
timeout 900 python bench.py > $O/bench_default.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_driver_like.log 2>&1
timeout 900 python bench.py --steps 938 --warmup 5 --no-extras > $O/bench_10s.log 2>&1
OW_TREM_TRAJ=0 timeout 900 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_per_instance_oscillators.log 2>&1
timeout 900 python bench.py --instances 65536 --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_65536.log 2>&1
timeout 900 python bench.py --instances 524288 --steps 12 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_524288.log 2>&1
timeout 900 python bench.py --host-rate 96000 --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_config3.log 2>&1
timeout 900 python bench.py --preamp melange --instances 65536 --steps 20 --warmup 3 --no-extras > $O/bench_melange.log 2>&1
timeout 900 python bench.py --power-amp melange --steps 4 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_power_amp_melange.log 2>&1
timeout 900 python bench.py --power-amp melange --instances 16384 --steps 4 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_power_amp_melange_16384.log 2>&1
timeout 900 python bench.py --workload batch > $O/bench_batch.log 2>&1
timeout 900 python bench.py --instances 256 --steps 100 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_p256.log 2>&1
# round 5: config 2 as the API delivers it (every block into the caller's pinned host block, ten re-strikes inside), the 8-rank host
# share priced on one GPU (two host threads: what LOCAL_WORLD_SIZE = 8 leaves a rank of a 16-CPU quota), a fresh 256-instance process
timeout 900 python bench.py --steps 938 --warmup 5 --no-extras --no-cpu-baseline --deliver host > $O/bench_10s_host.log 2>&1
OW_HOST_THREADS=2 timeout 900 python bench.py --steps 938 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_10s_2threads.log 2>&1
OW_HOST_THREADS=2 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_like_2threads.log 2>&1
timeout 900 python bench.py --instances 256 --steps 30 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_p256_fresh.log 2>&1
timeout 900 python tools/probe_fused.py > $O/probe_fused.log 2>&1
# the wave-level counters need the library built with -DOW_DBG_COUNTERS: rebuilt HERE from the round's sources (round 5 ran a stale one
# that lacked a symbol and recorded a traceback)
timeout 600 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -Wno-macro-redefined \
    -mllvm -amdgpu-sched-strategy=max-memory-clause -DOW_DBG_COUNTERS -o openwurli_amd/lib/libow_dbg.so openwurli_amd/csrc/openwurli_hip.hip > $O/build_dbg.log 2>&1
timeout 900 python tools/probe_power_amp_waves.py 16384 > $O/probe_power_amp_waves.log 2>&1
# round 6: the serial steps -- the trajectory's oscillator kernels alone (quad-lane vs row), the small-pool chain by pool size, a lone
# instance / 256 instances under the three chain kernels, the batch path under the row and the quad chain
timeout 900 python -m pytest tests/test_gpu_trajectory.py -q -s -k "row_oscillator" > $O/probe_oscillator_step.log 2>&1
timeout 900 python tools/probe_row_crossover.py > $O/probe_row_crossover.log 2>&1
timeout 900 python tools/bench_batch.py > $O/probe_batch_row.log 2>&1
OW_JOB_ROW=0 timeout 900 python tools/bench_batch.py > $O/probe_batch_quad.log 2>&1
OW_TREM_ROW=0 timeout 900 python bench.py --instances 256 --steps 30 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_p256_fresh_quad_oscillator.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -1 $O/bench_default.log | cut -c1-700
fi
ls $O
