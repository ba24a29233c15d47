#!/usr/bin/env python3
"""Melange 12-node preamp on the GPU against the oracle: literal per-sample rebuild (default) vs the rank-one kernel (OW_MEL_RANK1=1).
Prints the worst absolute deviation at the preamp node and at the output per scenario, and the kernel time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import openwurli_amd as ow
import oracle_binding as ob


def scenario(mode, name):
    if mode == "rank1":
        os.environ["OW_MEL_RANK1"] = "1"
    else:
        os.environ.pop("OW_MEL_RANK1", None)
    sr = 48000.0
    g = ow.EnginePool(sr, 2, preamp_kind=1)
    cs = [ob.OracleEngine(sr, preamp_kind=1) for _ in range(2)]
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)
    for k in range(2):
        for e in (g[k], cs[k]):
            e.set_volume(0.5); e.set_tremolo_depth(1.0 if name != "static" else 0.0)
            for n in (48, 60, 67) if k == 0 else (40, 72, 76, 91):
                e.note_on(n, 0.8)
    wp = wo = 0.0; peak_p = peak_o = 0.0
    for b in range(16):
        if name == "ramps" and b % 2 == 0:
            for k in range(2):
                for e in (g[k], cs[k]):
                    e.set_tremolo_depth(0.2 if (b // 2) % 2 else 1.0)       # depth-knob ramps: R_ldr moves fast
        go = g.render(512); gp = g.preamp_out(1024)
        for k in range(2):
            co, _, cp, _ = cs[k].render_taps(512)
            wp = max(wp, float(np.max(np.abs(gp[k] - cp)))); wo = max(wo, float(np.max(np.abs(go[k].astype(np.float64) - co))))
            peak_p = max(peak_p, float(np.max(np.abs(cp)))); peak_o = max(peak_o, float(np.max(np.abs(co))))
    g.close()
    print(f"{mode:8s} {name:8s}: preamp max|err| {wp:.3e} (peak {peak_p:.3f}), output max|err| {wo:.3e} (peak {peak_o:.3f})")


def timing(mode, n_eng=4096):
    if mode == "rank1":
        os.environ["OW_MEL_RANK1"] = "1"
    else:
        os.environ.pop("OW_MEL_RANK1", None)
    p = ow.EnginePool(48000.0, n_eng, preamp_kind=1)
    for k in range(0, n_eng, 7):
        p[k].note_on(60, 0.8)
    for _ in range(3):
        p.render(512, to_host=False)
    p.set_profiling(True)
    ms = []
    for _ in range(5):
        p.render(512, to_host=False); ms.append(p.last_kernel_ms()["preamp"])
    p.close()
    print(f"{mode:8s} k_preamp_mel* at {n_eng} engines: {np.mean(ms):.3f} ms per 512-sample block")


for mode in ("literal", "rank1"):
    for name in ("steady", "ramps", "static"):
        scenario(mode, name)
for mode in ("literal", "rank1"):
    timing(mode)
    timing(mode, 65536)
