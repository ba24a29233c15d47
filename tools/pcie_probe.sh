#!/bin/bash
for np in 1 2 4; do
  OW_PIPE=$np python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_pcie_$np.json 2> gpurun_out/r02_pcie_$np.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r02_pcie_$np.json"))
print("OW_PIPE=$np", "%.3e" % d["value"], "%.2f ms/step" % d["ms_per_step"], "pcie_inclusive %.3e %.2f ms" % (d["pcie_inclusive"]["value"], d["pcie_inclusive"]["ms_per_step"]))
PY
done
