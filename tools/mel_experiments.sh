# Round 6: the melange preamp's kernel (k_preamp_mel_col) under register-budget experiments, one library build each.
# usage (on the GPU box): bash tools/mel_experiments.sh <outdir>
O=${1:-gpurun_out/r06_mel}
mkdir -p $O
for lib in libopenwurli_hip.so libow_mel_w1.so; do
  [ -f openwurli_amd/lib/$lib ] || continue
  OPENWURLI_HIP_LIB=$PWD/openwurli_amd/lib/$lib timeout 900 python bench.py --preamp melange --instances 65536 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_$lib.log 2>&1
  echo "$lib $(tail -1 $O/bench_$lib.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", round(d["ms_per_step"],2), "preamp ms", round(d["roofline"]["kernel_ms_per_step"]["preamp"],2), "verified", d["verified"])')"
done | tee $O/summary.txt
