"""Scratch probe: cost of the blocks that follow a full-pool re-strike (run on the GPU box)."""
import time, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import openwurli_amd as ow, bench
n = 65536
pool = ow.EnginePool(48000.0, n)
s = bench.build_events(n, "strike"); r = bench.build_events(n, "restrike")
pool.midi(s)
for _ in range(8): pool.render(512, to_host=False)
pool.set_profiling(True)
t0 = time.perf_counter(); pool.midi(r); t1 = time.perf_counter()
print("restrike midi %.1f ms" % ((t1 - t0) * 1e3))
for b in range(6):
    t1 = time.perf_counter(); pool.render(512, to_host=False); t2 = time.perf_counter()
    print("block", b, "render %.1f ms" % ((t2 - t1) * 1e3), {k: round(v, 2) for k, v in pool.last_kernel_ms().items()}, flush=True)
