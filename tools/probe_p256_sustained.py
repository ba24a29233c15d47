"""BASELINE configs[4] literally (one fresh pool of 256 instances, no prefetch, no import) over 400 blocks of 512 samples: ms per block in
windows of 50, with and without the pool's per-kernel profiling events (bench.py's timed region has them on; they make a render wait for
the trajectory extension it launched for the blocks ahead).  usage (GPU box): python tools/probe_p256_sustained.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init(); torch.cuda.synchronize()          # (as bench.py: torch first)
import openwurli_amd as ow

def run(profiling, blocks=400, n_inst=256, sr=48000.0, buf=512):
    ow.binding.load_library().ow_test_clear_settle_caches()          # every process-wide store dropped: the pool starts the trajectory itself
    g = ow.EnginePool(sr, n_inst); g.set_sample_rate(sr)
    for k in range(n_inst):
        e = g[k]
        e.set_volume(0.5); e.set_tremolo_depth(0.5)
        for note in range(33, 97):
            e.note_on(note, (40 + (37 * k) % 88) / 127.0)
    g.set_profiling(profiling)
    win = []
    t0 = time.perf_counter()
    for b in range(blocks):
        g.render(buf, to_host=False)
        if (b + 1) % 50 == 0:
            t1 = time.perf_counter()
            win.append((t1 - t0) / 50 * 1e3)
            t0 = t1
    g.close()
    return win

def restrike(blocks_after=14, n_inst=256, sr=48000.0, buf=512):
    """... and what a whole-keyboard re-strike of the 256 instances costs (bench.py's config-2 script has one every 93.75 blocks, so its
    100-step line contains one and its 30-step line does not): render time of the blocks around it"""
    g = ow.EnginePool(sr, n_inst); g.set_sample_rate(sr)
    for k in range(n_inst):
        for note in range(33, 97):
            g[k].note_on(note, (40 + (37 * k) % 88) / 127.0)
    for _ in range(40):
        g.render(buf, to_host=False)
    for k in range(n_inst):
        for note in range(33, 97):
            g[k].note_off(note)
        for note in range(33, 97):
            g[k].note_on(note, (40 + (37 * k) % 88) / 127.0)
    t = []
    for _ in range(blocks_after):
        t0 = time.perf_counter(); g.render(buf, to_host=False); t.append((time.perf_counter() - t0) * 1e3)
    g.close()
    return t

for prof in (True, False):
    w = run(prof)
    print(f"profiling {'on ' if prof else 'off'}: ms per 512-sample block by window of 50: " + " ".join(f"{x:.2f}" for x in w) +
          f"; last window = {256 * 512 / (w[-1] * 1e-3) / 48000.0:.0f} x real time")
t = restrike()
print("blocks after a whole-keyboard re-strike of the 256 instances, ms each: " + " ".join(f"{x:.2f}" for x in t) + f"; sum over the first 10 minus 10 steady blocks = {sum(t[:10]) - 10 * t[-1]:.1f} ms")

# bench.py's own script (pool.midi bursts, buffers cut at the epoch, 256 decorrelated phases, profiling on): wall time of every step around
# the epoch boundary at sample 48 000 (block 93.75)
import bench
ow.binding.load_library().ow_test_clear_settle_caches()
g = ow.EnginePool(48000.0, 256); g.set_sample_rate(48000.0); g.ensure_buffer_capacity(512); g.stagger_tremolo(256)
sc = bench.Script(g, 256)
g.set_profiling(True)
ts = []
for i in range(125):
    t0 = time.perf_counter(); sc.step(profile=True); ts.append((time.perf_counter() - t0) * 1e3)
print("bench script, ms per step: steps 0-9 " + " ".join(f"{x:.2f}" for x in ts[:10]) + " | steps 85-110 " + " ".join(f"{x:.2f}" for x in ts[85:111]) +
      f" | mean 20-80 {np.mean(ts[20:80]):.2f}, mean 100-125 {np.mean(ts[100:125]):.2f}")
g.close()

# ... and what bench.py's closing barrier (torch.cuda.synchronize: every stream of the device, the trajectory store's included) adds
for n_steps in (35, 65, 105, 205):
    ow.binding.load_library().ow_test_clear_settle_caches()
    g = ow.EnginePool(48000.0, 256); g.set_sample_rate(48000.0); g.ensure_buffer_capacity(512); g.stagger_tremolo(256)
    sc = bench.Script(g, 256)
    for _ in range(5):
        sc.step()
    g.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_steps - 5):
        sc.step(profile=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n_steps - 5} timed steps: {1e3 * (t1 - t0) / (n_steps - 5):.3f} ms per step without the closing device-wide synchronise, which takes {1e3 * (t2 - t1):.1f} ms"
          f" -> {1e3 * (t2 - t0) / (n_steps - 5):.3f} ms per step with it; trajectory {g.trajectory_info()}")
    g.close()
