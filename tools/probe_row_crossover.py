"""Where the row-of-sixteen chain (k_chain_row: five wavefronts per eight engines) stops paying against the quad-lane one (k_chain_fused:
two per eight) and the lane-pair kernels: block time of a 512-sample block by pool size.  usage: python tools/probe_row_crossover.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import openwurli_amd as ow
sr, buf = 48000.0, 512
for n_eng in (256, 512, 1024, 2048, 4096):
    row = []
    for name, sw in (("row", {"chain_fused": 1, "chain_row": 1}), ("quad", {"chain_fused": 1, "chain_row": 0}), ("two launches, quad preamp", {"chain_fused": 0, "preamp_wide": 1}),
                     ("two launches, lane pairs", {"chain_fused": 0, "preamp_wide": 0})):
        p = ow.EnginePool(sr, n_eng)
        p.set_sample_rate(sr)
        for k, v in sw.items():
            p.set_switch(k, v)
        p.ensure_buffer_capacity(buf)
        p.stagger_tremolo(n_eng)
        ow.tremolo_prefetch(sr, 8.0)
        for k in range(n_eng):
            for n in range(40, 88, 2):
                p[k].note_on(n, 0.7)
        for _ in range(6): p.render(buf, to_host=False)
        t = time.perf_counter()
        for _ in range(20): p.render(buf, to_host=False)
        ms = 1e3 * (time.perf_counter() - t) / 20
        row.append("%s %.3f ms" % (name, ms))
        p.close()
    print("engines %5d: " % n_eng + "; ".join(row), flush=True)
