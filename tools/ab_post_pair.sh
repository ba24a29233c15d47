# A/B on one box: the oversampled output stage as lane pairs (k_post<true>) against lane = engine (k_post<false, true>), alternating runs.
# usage: bash tools/ab_post_pair.sh [instances ...]   (default: 131072)
for I in ${@:-131072}; do
for i in 1 2; do
for v in 0 1; do
OW_POST_PAIR=$v python bench.py --instances $I --steps 30 --warmup 5 --no-extras --no-cpu-baseline | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('instances $I post_pair $v', round(d['ms_per_step'],3), round(k['voices'],3), round(k['preamp'],3), round(k['post'],3), d['verified'])"
done; done; done
