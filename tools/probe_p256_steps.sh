# bench.py --instances 256 at several step counts in fresh processes, with the wall time of the last steps and of the closing barrier
for s in 30 100 100 200; do
OW_BENCH_STEP_MS=1 python bench.py --instances 256 --steps $s --warmup 5 --no-extras --no-cpu-baseline 2> /tmp/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $s', round(d['ms_per_step'],3), round(d['x_realtime_aggregate']))"
grep "step ms" /tmp/err.txt | python -c "
import sys
l=sys.stdin.read().split('steps: ')[1]
steps,bar=l.split(' | ')
v=[float(x) for x in steps.split()]
print('   max step', max(v), 'at', v.index(max(v)), '; sum', round(sum(v),1), 'ms; steps > 2 ms:', [(i,x) for i,x in enumerate(v) if x>2.0][:12], ';', bar.strip())"
done
