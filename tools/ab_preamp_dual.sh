# A/B on one box: k_preamp (lanes l, l + 32 = main, shadow) against k_preamp_dual (lane = engine), three alternating runs
for i in 1 2 3; do
for v in 0 1; do
OW_PREAMP_DUAL=$v python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('preamp_dual $v', round(d['ms_per_step'],3), round(k['voices'],3), round(k['preamp'],3), round(k['post'],3), d['verified'])"
done; done
