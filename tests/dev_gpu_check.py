"""Developer smoke script (not a pytest): quick GPU-vs-oracle numbers while bringing kernels up."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import openwurli_amd as ow
import oracle_binding as ob

t = time.time()
g = ow.render_note(60, 100 / 127.0, 0.25, 48000.0)
c = ob.render_note(60, 100 / 127.0, 0.25, 48000.0)
print("render_note", len(g), len(c), ob.parity_report(g, c), "t=%.2f" % (time.time() - t))

sr = 48000.0
t = time.time()
ge = ow.WurliEngine(sr)
print("engine new t=%.2f" % (time.time() - t)); t = time.time()
ge.set_sample_rate(sr)
print("set_sample_rate (warm-up) t=%.2f" % (time.time() - t))
ce = ob.OracleEngine(sr); ce.set_sample_rate(sr)
for e in (ge, ce):
    e.set_volume(0.5); e.set_tremolo_depth(0.5); e.set_speaker_character(0.0); e.set_mlp_enabled(True)
    for n in (48, 60, 64, 67, 84):
        e.note_on(n, 100 / 127.0)
gout = []; cout = []; gvs = []; cvs = []
pool_h = ge
for b in range(20):
    go = ge.render(512); gout.append(go.copy())
    co, cv = ce.render_tap(512); cout.append(co); cvs.append(cv)
gout = np.concatenate(gout); cout = np.concatenate(cout)
print("engine out", ob.parity_report(gout, cout))
print("gpu diag", ge.diag().active_voices, "cpu active", ce.active_voice_count())
