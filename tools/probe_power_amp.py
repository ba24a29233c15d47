#!/usr/bin/env python3
"""Melange power amp alone on the GPU (ow_debug_power_amp): rows x samples of a sine at a given level, wall time of the call.
usage: tools/probe_power_amp.py [rows] [samples] [amplitude]   (run it under rocprofv3 for the kernel's own time / counters)"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openwurli_amd import binding

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
amp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.001
L = binding.load_library()
sr = 96000.0
t = np.arange(n) / sr
x = np.ascontiguousarray(np.tile(amp * np.sin(2 * np.pi * 220 * t), (rows, 1)))
out = np.zeros_like(x); taps = np.zeros(x.shape + (3,))
for rep in range(2):
    t0 = time.time()
    rc = L.ow_debug_power_amp(C.c_double(sr), x.ctypes.data_as(C.c_void_p), C.c_size_t(rows), C.c_size_t(n), 1, None, None, None,
                              out.ctypes.data_as(C.c_void_p), taps.ctypes.data_as(C.c_void_p), 0)
    dt = time.time() - t0
    print(f"rc {rc} rows {rows} n {n} amp {amp}: {dt * 1e3:.1f} ms wall, newton iterations mean {taps[0, :, 0].mean() + 1:.2f} max {taps[0, :, 0].max() + 1:.0f}, "
          f"{rows * n / dt / 1e6:.2f} M chain samples/s, peak out {np.abs(out).max():.4f}")

if len(sys.argv) > 4 and sys.argv[4] == "chord":
    # the bench's config-2 chord: PA input = 0.25 x preamp output of the first blocks, per-instance velocities
    sys.path.insert(0, ROOT)
    import bench
    import openwurli_amd as ow
    ne = 64
    g = ow.EnginePool(48000.0, ne)
    if len(sys.argv) > 5 and sys.argv[5] == "init":
        g.set_sample_rate(48000.0)          # the plugin's initialize(), as bench.py does
    sc = bench.Script(g, ne)
    rows_in = []
    for b in range(4):
        sc.step()
        rows_in.append(g.preamp_out(1024).copy())
    xin = np.ascontiguousarray(np.concatenate(rows_in, axis=1) * 0.25)
    o = np.zeros_like(xin); tp = np.zeros(xin.shape + (3,))
    for rep in range(2):
        t0 = time.time()
        rc = L.ow_debug_power_amp(C.c_double(96000.0), xin.ctypes.data_as(C.c_void_p), C.c_size_t(ne), C.c_size_t(xin.shape[1]), 1, None, None, None,
                                  o.ctypes.data_as(C.c_void_p), tp.ctypes.data_as(C.c_void_p), 0)
        dt = time.time() - t0
    it = tp[:, :, 0] + 1
    print(f"chord: input peak {np.abs(xin).max():.4f}; {dt * 1e3:.1f} ms for {xin.shape[1]} samples ({dt / xin.shape[1] * 1e6:.1f} us/sample); "
          f"solves/sample mean {it.mean():.2f}, per-sample max over 8-engine groups mean {it.reshape(8, 8, -1).max(axis=1).mean():.2f}, "
          f"over all 64 {it.max(axis=0).mean():.2f}; max {it.max():.0f}; guard resets {tp[:, -1, 1].max():.0f}")
    print("  histogram of solves per sample (1,2,3,4,5-8,9-16,17-32,33-70,71):", np.histogram(it, bins=[1, 2, 3, 4, 5, 9, 17, 33, 71, 72])[0])
    print("  per-engine mean solves/sample: min %.2f max %.2f" % (it.mean(axis=1).min(), it.mean(axis=1).max()))
    for b in range(4):
        seg = it[:, b * 1024:(b + 1) * 1024]
        print(f"  block {b}: mean {seg.mean():.2f}, group-max mean {seg.reshape(8, 8, -1).max(axis=1).mean():.2f}, max {seg.max():.0f}")

if len(sys.argv) > 4 and sys.argv[4] == "pool":
    import bench
    import openwurli_amd as ow
    for ne in (32, 256, 8192, 16384):
        g = ow.EnginePool(48000.0, ne, power_amp_kind=1)
        g.set_profiling(True)
        sc = bench.Script(g, ne)
        for b in range(3):
            sc.step(profile=True)
            ms = g.last_kernel_ms()
            d = g[ne - 1].power_amp_diag()
            print(f"pool {ne}: block {b}: post {ms['post']:.1f} ms; engine {ne - 1} diag nr_max {d.nr_max_iter_count} guard {d.guard_resets} peak {d.peak_output_volts:.4f}")
        g.close()

if len(sys.argv) > 4 and sys.argv[4] == "pool2":
    import bench
    import openwurli_amd as ow
    ne = 16384
    for variant in ("plain", "set_sample_rate", "setters", "both"):
        g = ow.EnginePool(48000.0, ne, power_amp_kind=1)
        if variant in ("set_sample_rate", "both"):
            g.set_sample_rate(48000.0)
        if variant in ("setters", "both"):
            for k in range(4096):
                e = g[k]
                e.set_volume(0.5); e.set_tremolo_depth(0.5); e.set_speaker_character(0.0); e.set_mlp_enabled(True)
        g.set_profiling(True)
        sc = bench.Script(g, ne)
        for b in range(2):
            sc.step(profile=True)
            ms = g.last_kernel_ms()
            d = g[5].power_amp_diag()
            print(f"{variant}: block {b}: post {ms['post']:.1f} ms; engine 5 diag nr_max {d.nr_max_iter_count} guard {d.guard_resets} peak {d.peak_output_volts:.4f}")
        g.close()

if len(sys.argv) > 4 and sys.argv[4] == "diag":
    import bench
    import openwurli_amd as ow
    ne = 64
    g = ow.EnginePool(48000.0, ne, power_amp_kind=1)
    g.set_sample_rate(48000.0)
    g.set_profiling(True)
    sc = bench.Script(g, ne)
    for b in range(8):
        sc.step(profile=True)
        ds = [g[k].power_amp_diag() for k in range(ne)]
        print(f"block {b}: post {g.last_kernel_ms()['post']:.1f} ms; BE retries so far: total {sum(d.nr_max_iter_count for d in ds)} (max per engine {max(d.nr_max_iter_count for d in ds)}), "
              f"guard resets total {sum(d.guard_resets for d in ds)}, nan resets {sum(d.nan_resets for d in ds)}")
    g.close()
