# The reference's rounding discipline, priced: the library built with -ffp-contract=fast everywhere (openwurli_amd/lib/libow_fma.so;
# `OW_FP_CONTRACT=fast bash build.sh` builds the product that way) next to the default build (contraction only inside the voice step),
# same box, same commands; then the parity tests on the contracted build (which floors it passes is the point).
# usage (GPU box): bash tools/fma_mode.sh
O=gpurun_out/fma; mkdir -p $O
for v in default fma; do
  if [ $v = fma ]; then export OPENWURLI_HIP_LIB=$PWD/openwurli_amd/lib/libow_fma.so; else unset OPENWURLI_HIP_LIB; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  timeout 600 python bench.py --preamp melange --instances 65536 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/melange_$v.json 2> $O/melange_$v.err
  timeout 900 python bench.py --power-amp melange --instances 65536 --steps 2 --warmup 3 --no-extras --no-cpu-baseline > $O/mpa_$v.json 2> $O/mpa_$v.err
  python - <<PY
import json
for n in ("bench", "melange", "mpa"):
    d = json.load(open("$O/%s_$v.json" % n)); k = d["roofline"]["kernel_ms_per_step"]
    print("$v", n, "%.4g samples/s" % d["value"], "ms/step %.2f" % d["ms_per_step"], {a: round(b, 2) for a, b in k.items()}, "verified", d["verified"])
PY
done
export OPENWURLI_HIP_LIB=$PWD/openwurli_amd/lib/libow_fma.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_power_amp.py tests/test_gpu_trajectory.py -q -m gpu 2>&1 | tail -25 > $O/parity_fma.txt
tail -12 $O/parity_fma.txt
