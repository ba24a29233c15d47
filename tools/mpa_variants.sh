for v in old trips4 trips6 trips8 trips16; do
  if [ $v = trips4 ]; then unset OPENWURLI_HIP_LIB; else export OPENWURLI_HIP_LIB=$PWD/openwurli_amd/lib/variants/$v.so; fi
  python bench.py --power-amp melange --instances 65536 --steps 2 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/mpa_$v.json 2>gpurun_out/mpa_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/mpa_$v.json')); print('$v', round(d['value']/1e6,2), 'Msamples/s post', round(d['roofline']['kernel_ms_per_step']['post'],1), 'ms passes', round(d['roofline']['power_amp_newton_passes_per_chain_sample'],3))"
done
