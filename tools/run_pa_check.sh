python -m pytest tests/test_gpu_power_amp.py tests/test_gpu_division.py tests/test_gpu_render_flags.py -x -q -m gpu 2>&1 | tail -5
python tools/probe_power_amp.py 8192 512 0.001
python tools/probe_power_amp.py 8192 512 0.3
python bench.py --power-amp melange --instances 16384 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --tremolo-groups 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mpa 16384 value %.3e ms/step %.2f'%(d['value'],d['ms_per_step']), d['roofline']['kernel_ms_per_step'])
"
