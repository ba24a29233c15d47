#!/bin/bash
# staged render: stage-count sweep on the default bench workload (one GPU)
for np in 1 2 4 8; do
  OW_PIPE=$np python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r02_pipe_$np.json 2> gpurun_out/r02_pipe_$np.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r02_pipe_$np.json"))
print("OW_PIPE=$np", "%.3e samples/s" % d["value"], "%.2f ms/step" % d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])
PY
done
