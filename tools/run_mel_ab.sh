python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "melange" 2>&1 | tail -5
python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "setter" 2>&1 | tail -3
for mode in col lds; do
  for g in 1 0; do
    if [ $mode = lds ]; then export OW_MEL_LDS=1; else unset OW_MEL_LDS; fi
    python bench.py --preamp melange --instances 65536 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --tremolo-groups $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode groups=$g value %.3e ms/step %.2f'%(d['value'],d['ms_per_step']), d['roofline']['kernel_ms_per_step'])
"
  done
done
