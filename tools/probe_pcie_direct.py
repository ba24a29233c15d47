"""Steady blocks of the bench pool rendered into a pinned host block: staged copy (d_out -> host after the last kernel) against the
output stage storing straight into the mapped block (`out_direct`).  usage: python3 tools/probe_pcie_direct.py [instances]"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
import openwurli_amd as ow

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
torch.cuda.set_device(0)
p = ow.EnginePool(48000.0, n, device=0)
p.set_sample_rate(48000.0)
p.ensure_buffer_capacity(bench.BUF)
p.stagger_tremolo(n)
sc = bench.Script(p, n)
for _ in range(12):
    sc.step()
host = p.alloc_host_block(bench.BUF)
res = {}
for name, sw in (("hbm", None), ("staged", 0), ("direct", 1), ("staged2", 0), ("direct2", 1)):
    s2 = bench.Script(p, n, host_out=None if sw is None else host)
    s2.pos = sc.pos
    if sw is not None:
        p.set_switch("out_direct", sw)
    p.set_profiling(True)
    for _ in range(3):
        s2.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 12
    s2.kernel_ms[:] = 0; s2.kernel_launches = 0
    for _ in range(K):
        s2.step(profile=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    p.set_profiling(False)
    km = s2.kernel_ms / max(s2.kernel_launches, 1)
    res[name] = {"ms_per_step": 1e3 * el / K, "samples_per_s": K * bench.BUF * n / el, "kernel_ms": [float(x) for x in km]}
    sc.pos = s2.pos
    if sw == 1:      # the block the host received equals the block left in HBM
        a = np.ctypeslib.as_array((__import__("ctypes").c_float * (n * bench.BUF)).from_address(host[0])).reshape(n, bench.BUF)
        res[name]["host_equals_hbm"] = bool(np.array_equal(a, p.last_block()))
print(json.dumps(res, indent=1))
p.free_host_block(host)
p.close()
