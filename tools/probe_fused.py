import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import openwurli_amd as ow
sr = 48000.0
for fused, row in ((0, 0), (1, 0), (1, 1)):
    for n_eng, buf in ((1, 64), (1, 512), (256, 512)):
        p = ow.EnginePool(sr, n_eng)
        p.set_sample_rate(sr)
        p.set_switch("chain_fused", fused); p.set_switch("chain_row", row)
        p.ensure_buffer_capacity(buf)
        if n_eng > 1: p.stagger_tremolo(n_eng)
        ow.tremolo_prefetch(sr, 8.0)
        for k in range(n_eng):
            for n in range(33, 97):
                p[k].note_on(n, 0.7)
        for _ in range(8): p.render(buf, to_host=(n_eng == 1))
        lat = []
        for _ in range(60):
            t = time.perf_counter(); p.render(buf, to_host=(n_eng == 1)); lat.append(time.perf_counter() - t)
        p.set_profiling(True)
        ms = []
        for _ in range(10):
            p.render(buf, to_host=(n_eng == 1)); ms.append(list(p.last_kernel_ms().values()))
        p.set_profiling(False)
        ms = np.mean(np.array(ms), axis=0)
        print(f"fused {fused} row {row} engines {n_eng} buffer {buf}: wall/buffer {1e6*np.mean(lat):.0f} us (p50 {1e6*np.median(lat):.0f}); x real time {n_eng*buf/np.mean(lat)/sr:.1f}; kernels ms ops/voices/tremolo/preamp/post = {np.round(ms,3).tolist()}")
        p.close()
