"""Developer script: locate melange deviations at depth 1.0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import openwurli_amd as ow
import oracle_binding as ob
sr = 48000.0
g = ow.EnginePool(sr, 1, preamp_kind=1); c = ob.OracleEngine(sr, preamp_kind=1)
g.set_sample_rate(sr); c.set_sample_rate(sr)
for e in (g[0], c):
    e.set_volume(0.5); e.set_tremolo_depth(1.0); e.set_speaker_character(0.0)
    for n in (40, 72, 76, 91): e.note_on(n, 0.8)
GP=[];CP=[];CR=[]
for b in range(16):
    g.render(512); gp = g.preamp_out(1024)[0]
    co,_,cp,cr = c.render_taps(512)
    GP.append(gp);CP.append(cp);CR.append(cr)
GP,CP,CR = map(np.concatenate,(GP,CP,CR))
d = np.abs(GP-CP)
print("max", d.max(), "at", d.argmax(), "count>1e-8", (d>1e-8).sum())
idx = np.nonzero(d>1e-8)[0]
if idx.size:
    i0 = idx[0]
    print("first region", i0, "R there", CR[i0-2:i0+6])
    print("err", (GP-CP)[i0-3:i0+12])
    print("cpu", CP[i0-3:i0+12])
    # cluster starts
    starts = idx[np.insert(np.diff(idx)>50, 0, True)]
    print("cluster starts", starts[:20], "R at starts", CR[starts[:20]])
print("Rmin/max", CR.min(), CR.max())
