"""Developer script: melange preamp GPU-vs-oracle deviation sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import openwurli_amd as ow
import oracle_binding as ob
sr = 48000.0
t=time.time(); g = ow.EnginePool(sr, 1, preamp_kind=1); print("pool new", time.time()-t)
c = ob.OracleEngine(sr, preamp_kind=1)
t=time.time(); g.set_sample_rate(sr); print("warm", time.time()-t); c.set_sample_rate(sr)
for e in (g[0], c):
    e.set_tremolo_depth(0.5)
    for n in (48, 60, 67): e.note_on(n, 0.8)
GP=[];CP=[];GO=[];CO=[]
t=time.time()
for b in range(40):
    go = g.render(512)[0]; gp = g.preamp_out(1024)[0]
    co,_,cp,_ = c.render_taps(512)
    GP.append(gp);CP.append(cp);GO.append(go);CO.append(co)
print("render t", time.time()-t)
GP,CP,GO,CO = map(np.concatenate,(GP,CP,GO,CO))
print("preamp", ob.parity_report(GP,CP)); print("out", ob.parity_report(GO,CO))
print("diag", g[0].diag().preamp_nan_resets, g[0].diag().tremolo_be_fallbacks)
