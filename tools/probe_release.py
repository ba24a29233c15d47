"""Played input with releases: per-block voice time of a pool in which every engine holds a chord and releases / re-plays part of it,
with the release variant of the steady voice kernel on and off.  Usage: python tools/probe_release.py [engines]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import openwurli_amd as ow

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dt = np.dtype(ow.binding.MIDI_DTYPE)

def events(kind, notes, vel=0.7):
    ev = np.zeros(n * len(notes), dtype=dt)
    ev["engine"] = np.repeat(np.arange(n, dtype=np.uint32), len(notes))
    ev["type"] = kind
    ev["note"] = np.tile(np.array(notes, dtype=np.uint8), n)
    ev["value"] = vel
    return ev

for release in (1, 0):
    p = ow.EnginePool(48000.0, n)
    p.set_sample_rate(48000.0)
    p.set_switch("voice_release", release)
    p.set_profiling(True)
    chord = [36, 43, 48, 52, 55, 60, 64, 67, 72, 76, 79, 84]
    p.midi(events(0, chord))
    for b in range(8):
        p.render(512, to_host=False)
        print("  strike block", b, p.get_switch("blocks_attack"), p.get_switch("blocks_general"), p.get_switch("blocks_steady"), flush=True)
    p.midi(events(1, chord[::2]))                # half of the keys up: six voices damp beside six held ones
    ms = []
    for b in range(10):
        p.render(512, to_host=False)
        ms.append(p.last_kernel_ms()["voices"])
        print("  release block", b, p.get_switch("blocks_attack"), p.get_switch("blocks_general"), p.get_switch("blocks_steady"), p[0].active_voice_count(), flush=True)
    print("release variant" if release else "general kernel ", n, "engines x 12 voices, 6 damping: voices ms per block", [round(float(x), 2) for x in ms])
    p.close()
