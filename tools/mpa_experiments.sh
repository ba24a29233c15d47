# Round 6: the melange power amp's kernel (k_post_mpa) under the experiments VERDICT r05 item 2 names, one library build each
# (openwurli_amd/lib/libow_pa_<name>.so, built by the caller with the flags listed in profiles/r06_mpa_experiments.md).
# usage (on the GPU box): bash tools/mpa_experiments.sh <outdir>
O=${1:-gpurun_out/r06_mpa}
mkdir -p $O
for lib in libopenwurli_hip.so libow_pa_t8.so libow_pa_w8.so libow_pa_ui.so; do
  [ -f openwurli_amd/lib/$lib ] || continue
  OPENWURLI_HIP_LIB=$PWD/openwurli_amd/lib/$lib timeout 900 python bench.py --power-amp melange --instances 16384 --steps 4 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_$lib.log 2>&1
  echo "$lib $(tail -1 $O/bench_$lib.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", round(d["ms_per_step"],1), "post ms", round(d["roofline"]["kernel_ms_per_step"]["post"],1), "verified", d["verified"])')"
done | tee $O/summary.txt
OPENWURLI_HIP_LIB=$PWD/openwurli_amd/lib/libow_pa_t8.so timeout 900 python -m pytest tests/test_gpu_power_amp.py -x -q 2>&1 | tail -2 | tee $O/tests_t8.txt
[ -f openwurli_amd/lib/libow_pa_ui.so ] && OPENWURLI_HIP_LIB=$PWD/openwurli_amd/lib/libow_pa_ui.so timeout 900 python -m pytest tests/test_gpu_power_amp.py -x -q 2>&1 | tail -2 | tee $O/tests_ui.txt
