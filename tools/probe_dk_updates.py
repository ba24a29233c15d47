"""Newton updates of dk_step on the bench chord, by kind of solver state (development counters: needs a library built with
-DOW_DBG_COUNTERS as openwurli_amd/lib/libow_dbg.so).  k_preamp puts 32 main states in lanes 0-31 and their 32 shadow states in lanes
32-63 of a wavefront and runs the Newton loop wave-uniform: how many updates does a state need itself, the slowest state of each half,
and how many does the wavefront execute?  usage: tools/probe_dk_updates.py [engines]"""
import os, sys, ctypes as C
os.environ["OPENWURLI_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openwurli_amd", "lib", "libow_dbg.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openwurli_amd as ow
from openwurli_amd import binding
import bench
lib = binding.load_library()
out = (C.c_ulonglong * 8)()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = ow.EnginePool(48000.0, n)
p.set_sample_rate(48000.0)
p.ensure_buffer_capacity(512)
p.stagger_tremolo(n)
p.set_switch("preamp_wide", 0); p.set_switch("chain_fused", 0)
sc = bench.Script(p, n)
for _ in range(6): sc.step()
lib.ow_debug_counters(out, 0)
blocks = 4
for _ in range(blocks): sc.step()
lib.ow_debug_counters(out, 0)
ws = out[3]                      # wavefront-samples (dk_step calls per wavefront)
print(f"engines {n}, {blocks} blocks: per chain sample -- updates a main state needs {out[0] / (32 * ws):.3f}, a shadow state {out[1] / (32 * ws):.3f}; "
      f"the slowest of a wavefront's 32 mains {out[4] / ws:.3f}, of its 32 shadows {out[5] / ws:.3f}; updates the wavefront executes {out[2] / ws:.3f}")
