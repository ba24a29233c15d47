for ser in 0 1; do
  OW_TREM_SERIAL=$ser python bench.py --steps 30 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('serial=$ser value %.3e ms/step %.2f'%(d['value'],d['ms_per_step']), d['roofline']['kernel_ms_per_step'], 'frac', d['roofline']['frac'])
"
done
python bench.py --steps 30 --warmup 4 --no-extras --no-cpu-baseline --tremolo-groups 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('shared value %.3e ms/step %.2f'%(d['value'],d['ms_per_step']), d['roofline']['kernel_ms_per_step'], 'frac', d['roofline']['frac'])
"
python bench.py --instances 256 --steps 30 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('p256 value %.3e ms/step %.2f'%(d['value'],d['ms_per_step']), d['roofline']['kernel_ms_per_step'])
"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_tremolo_groups.py -q -m gpu -k "tremolo or steady or wide" 2>&1 | tail -3
