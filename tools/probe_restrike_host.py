"""Where the host time of a whole-pool re-strike goes: ow_pool_midi (16.7 M events) and the render that follows it (op packing, upload,
voice lists, k_apply_ops, general voice kernel).  python tools/probe_restrike_host.py [instances]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
import openwurli_amd as ow

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
p = ow.EnginePool(48000.0, n)
p.set_sample_rate(48000.0)
strike = bench.build_events(n, "strike"); restrike = bench.build_events(n, "restrike")
p.midi(strike)
for _ in range(6):
    p.render(512)
for rep in range(3):
    t0 = time.perf_counter(); p.midi(restrike); t1 = time.perf_counter()
    p.render(128); t2 = time.perf_counter()
    ms = p.last_kernel_ms() if hasattr(p, "last_kernel_ms") else None
    p.render(512); t3 = time.perf_counter()
    for _ in range(4):
        p.render(512)
    t4 = time.perf_counter(); p.render(512); t5 = time.perf_counter()
    print("re-strike %d: midi %.1f ms, first block (128) %.1f ms, next (512) %.1f ms, steady block %.1f ms" % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t5 - t4)), ms)
p.close()
