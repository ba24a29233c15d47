"""Developer script: stage-wise taps, GPU vs oracle, on the smoke scenario."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import openwurli_amd as ow
import oracle_binding as ob
sr = 48000.0
spk = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
g = ow.EnginePool(sr, 1); c = ob.OracleEngine(sr)
for e in (g[0], c):
    e.set_volume(0.5); e.set_tremolo_depth(0.5); e.set_speaker_character(spk); e.set_mlp_enabled(True)
    for n in (45, 60, 64, 79):
        e.note_on(n, 0.8)
GO=[];CO=[];GV=[];CV=[];GP=[];CP=[]
for b in range(8):
    go = g.render(256)[0]; gv = g.voice_sum(256)[0]; gp = g.preamp_out(512)[0]
    co, cv, cp, cr = c.render_taps(256)
    GO.append(go);CO.append(co);GV.append(gv);CV.append(cv);GP.append(gp);CP.append(cp)
GO,CO,GV,CV,GP,CP = map(np.concatenate,(GO,CO,GV,CV,GP,CP))
print("voice sum ", ob.parity_report(GV, CV))
print("preamp out", ob.parity_report(GP, CP))
print("output    ", ob.parity_report(GO, CO))
d = GP-CP; i = np.argmax(np.abs(d)); print("preamp worst idx", i, d[max(0,i-3):i+4], CP[max(0,i-3):i+4])
d = GV-CV; i = np.argmax(np.abs(d)); print("voice worst idx", i, d[max(0,i-3):i+4], CV[max(0,i-3):i+4])
