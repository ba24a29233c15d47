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
# (name, deliver to the host block?, chain_stream switch, out_direct switch)
modes = (("hbm_two_launches", False, 0, 0), ("hbm_stream_kernel", False, 1, 0), ("host_staged_copy", True, 0, 0), ("host_k_post_direct", True, 0, 1),
         ("host_stream_kernel", True, 1, 1), ("hbm_two_launches_again", False, 0, 0), ("host_stream_kernel_again", True, 1, 1))
if len(sys.argv) > 2 and sys.argv[2] == "general":
    modes = (("steady_kernel", False, 0, 0), ("general_kernel_no_phase_active", False, 0, 0))
for name, to_host, cs, od in modes:
    s2 = bench.Script(p, n, host_out=host if to_host else None)
    s2.pos = sc.pos
    p.set_switch("chain_stream", cs); p.set_switch("out_direct", od)
    if name.startswith("general"):
        p.set_switch("force_general", 1)
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
    res[name] = {"ms_per_step": 1e3 * el / K, "samples_per_s": K * bench.BUF * n / el, "kernel_ms": dict(zip(("ops", "voices", "tremolo", "preamp", "post"), (float(x) for x in km)))}
    sc.pos = s2.pos
    if to_host:      # the block the host received equals the block left in HBM
        a = np.ctypeslib.as_array((__import__("ctypes").c_float * (n * bench.BUF)).from_address(host[0])).reshape(n, bench.BUF)
        res[name]["host_equals_hbm"] = bool(np.array_equal(a, p.last_block()))
print(json.dumps(res, indent=1))
p.free_host_block(host)
p.close()
