# Kernel timeline of one whole-pool re-strike (131 072 instances x 64 keys): `gpurun -- bash tools/restrike_trace.sh`, then
# `python3 tools/restrike_trace.py > profiles/rNN_restrike_trace.txt` here.  bench.py's epoch is 94 blocks: 88 warm-up steps put the
# re-strike into the 10 timed ones.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/restrike; mkdir -p $O
OW_HOST_PROFILE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 10 --warmup 88 --no-extras --no-cpu-baseline ${OW_TRACE_ARGS} > $O/log.txt 2>&1
python3 tools/restrike_trace.py
grep hostprof $O/log.txt | tail -3
