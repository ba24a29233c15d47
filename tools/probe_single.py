#!/usr/bin/env python3
"""Where does the time of ONE engine go?  Per-kernel HIP-event times and wall time per buffer for a pool of one (64 voices)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import openwurli_amd as ow

def main():
    sr = 48000.0
    for buf in (64, 128, 256, 512):
        p = ow.EnginePool(sr, 1)
        p.set_sample_rate(sr)
        p.ensure_buffer_capacity(buf)
        for n in range(33, 97):
            p[0].note_on(n, 0.7)
        for _ in range(6):
            p.render(buf)
        lat = []
        for _ in range(40):
            t = time.perf_counter(); p.render(buf); lat.append(time.perf_counter() - t)
        p.set_profiling(True)
        ms = []
        for _ in range(10):
            p.render(buf); ms.append(list(p.last_kernel_ms().values()))
        p.set_profiling(False)
        ms = np.mean(np.array(ms), axis=0)
        print(f"buffer {buf}: wall/buffer {1e6*np.mean(lat):.0f} us (p50 {1e6*np.median(lat):.0f}) = {buf/np.mean(lat):.3g} samples/s; kernels ms ops/voices/tremolo/preamp/post = {np.round(ms,3).tolist()}")
        p.close()

if __name__ == "__main__":
    main()
