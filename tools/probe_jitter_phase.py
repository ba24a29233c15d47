"""Steady voice kernel against the alignment of the voices' 16-sample jitter grid (reed.rs:262: every voice updates its OU jitter when
ITS sample counter is a multiple of 16).  The bench strikes all 64 keys of an engine at one sample, so all lanes of a wavefront update
together; played input does not.  python tools/probe_jitter_phase.py [instances]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
import openwurli_amd as ow
from openwurli_amd import binding

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
BL = int(sys.argv[2]) if len(sys.argv) > 2 else 512
for mode in ("aligned", "staggered"):
    p = ow.EnginePool(48000.0, n)
    p.set_sample_rate(48000.0)
    p.set_profiling(True)
    notes = list(range(33, 97))
    if mode == "aligned":
        p.midi(bench.build_events(n, "strike"))
        pass
    else:
        for i, m in enumerate(notes):       # one key per 1-sample block: 64 different phases of the 16-sample grid in every engine
            ev = np.zeros(n, dtype=np.dtype(binding.MIDI_DTYPE))
            ev["engine"] = np.arange(n, dtype=np.uint32); ev["type"] = 0; ev["note"] = m; ev["value"] = 0.8
            p.midi(ev)
            p.render(1, to_host=False)
    for _ in range(6):                      # past the onset ramps and the attack noise
        p.render(BL, to_host=False)
    ms = []
    for _ in range(10):
        p.render(BL, to_host=False)
        ms.append(p.last_kernel_ms()["voices"])
    print("%-10s %d instances: k_voice_steady %.2f ms per %d-sample block" % (mode, n, float(np.median(ms)), BL))
    p.close()
