"""How often a wavefront of k_post_mpa executes the pieces of the melange power-amp solver on the bench chord (development counters:
needs a library built with -DOW_DBG_COUNTERS as openwurli_amd/lib/libow_dbg.so).  usage: tools/probe_power_amp_waves.py [engines]"""
import os, sys, ctypes as C
os.environ["OPENWURLI_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openwurli_amd", "lib", "libow_dbg.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openwurli_amd as ow
from openwurli_amd import binding
import bench
lib = binding.load_library()
out = (C.c_ulonglong * 8)()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
p = ow.EnginePool(48000.0, n, power_amp_kind=1)
p.set_sample_rate(48000.0)
p.ensure_buffer_capacity(512)
p.stagger_tremolo(n)
sc = bench.Script(p, n)
for _ in range(4): sc.step()
lib.ow_debug_counters(out, 0)
blocks = 2
for _ in range(blocks): sc.step()
lib.ow_debug_counters(out, 0)
waves = n // 8
ws = blocks * 1024 * waves          # wavefront-samples
print(f"engines {n}: per wavefront and chain-rate sample: main-loop trips {out[2] / ws:.2f}, passes of any kind {out[7] / ws:.2f} (BE retry {out[4] / ws:.3f}), "
      f"device evaluations {out[3] / ws:.1f} = {out[3] / max(out[7], 1):.2f} per pass, sample completions {out[6] / ws:.2f}; "
      f"engine passes per sample {p.power_amp_passes().mean() / 1024:.2f}")
