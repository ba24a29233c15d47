"""Newton sweeps of the melange preamp as the device executes them: per lane and per wavefront, for a pool that shares one tremolo phase
and for one oscillator per engine.  Needs a library built with the development counters:
    OW_HIPCC_EXTRA=-DOW_DBG_COUNTERS hipcc ... -o openwurli_amd/lib/libow_dbg.so   (the flags of build.sh)
Measured (round 3, 4 096 engines, six blocks of 512 after the warm-up): shared phase 1.35 sweeps per lane and sample, 2.10 per wavefront;
one oscillator per engine 3.86 / 3.94 -- the sweep count follows the tremolo phase (R moving fast costs sweeps), a decorrelated pool
averages over all phases and its wavefronts lose nothing to lock step."""
import os, sys, ctypes as C
os.environ["OPENWURLI_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openwurli_amd", "lib", "libow_dbg.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openwurli_amd as ow
from openwurli_amd import binding
import bench
lib = binding.load_library()
out = (C.c_ulonglong * 8)()
for groups in (1, 0):
    n = 4096
    p = ow.EnginePool(48000.0, n, preamp_kind=1)
    p.set_sample_rate(48000.0)
    p.ensure_buffer_capacity(512)
    if groups == 0:
        p.stagger_tremolo(n)
    sc = bench.Script(p, n)
    for _ in range(4): sc.step()
    lib.ow_debug_counters(out, 0)
    for _ in range(6): sc.step()
    lib.ow_debug_counters(out, 0)
    lane_bodies, wave_bodies = out[0], out[1]
    samples = 6 * 512 * 2
    print("groups", groups or n, "sweeps per lane-sample %.3f" % (lane_bodies / (samples * n * 2)), "per wave-sample %.3f" % (wave_bodies / (samples * (n // 32))))
    p.close()
