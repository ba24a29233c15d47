"""The shared tremolo trajectory (openwurli_hip.hip `TremTraj`): Tremolo::process takes no audio and no depth into the oscillator, the
LED envelope or r_ldr (tremolo.rs:121-146; depth only enters shunt_impedance, :152-167) and new() / reset() leave the same settled state
(:83-102, :192-216), so r_ldr[t] is ONE sequence per chain rate.  The library keeps it in HBM once per (device, chain rate) and every
engine reads it at its own t.  That must be invisible: a pool on the trajectory and a pool with one oscillator per phase group
(OW_TREM_TRAJ=0 at creation: rounds 1-3) produce the same bits -- R rows, preamp tap and output -- under random offsets, mid-run reset,
set_sample_rate, warm-up of one engine and depth ramps; both match the oracle; engines that outlive the store's capacity continue on their
own oscillator without a seam; pools of one created at different times share one store."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _pool(ow, sr, n, traj, **kw):
    with _Env(OW_TREM_TRAJ=None if traj else "0"):
        p = ow.EnginePool(sr, n, **kw)
    assert p.get_switch("trem_traj") == (1 if traj else 0)
    return p


def _script(sr, n, groups, blocks):
    """The same calls on both pools (and on oracle engines for a few k): returns per block (out, R rows, preamp tap)."""
    def run(p, cs=None, step=0):
        osr = 2 if sr < 88200.0 else 1
        res = []
        targets = [(k, p[k]) for k in range(n)] if cs is None else [(k, c) for k, c in cs.items()]
        for k, e in targets:
            e.set_tremolo_depth(0.15 + 0.8 * ((7 * k) % 10) / 10.0); e.set_volume(0.4 + 0.01 * (k % 9))
            e.note_on(40 + (5 * k) % 50, 0.55 + 0.04 * (k % 10)); e.note_on(64 + k % 12, 0.8)
        for b in range(blocks):
            length = (512, 300, 512, 64, 512, 1, 512, 777)[b % 8]
            if b == 3:
                for k, e in targets:
                    if k % 3 == 0:
                        e.set_tremolo_depth(1.0)                      # depth ramp
            if b == 5:
                for k, e in targets:
                    if k == 2:
                        e.reset(); e.note_on(70, 0.7)                 # one engine starts over at t = 0
            if b == 8:
                for k, e in targets:
                    if k == 4:
                        e.warm_up()                                   # 0.6 s on one engine only: it runs ahead of the pool clock
            if cs is None:
                out = p.render(length)
                res.append((out.copy(), p.tremolo_r(length * osr).copy(), p.preamp_out(length * osr).copy()))
            else:
                res.append({k: c.render_taps(length, osr) for k, c in cs.items()})
        return res
    return run


@pytest.mark.parametrize("sr", [48000.0, 96000.0])
def test_trajectory_is_bit_identical_to_per_group_oscillators(hiplib, oracle, sr):
    import openwurli_amd as ow
    n, groups, blocks = 70, 23, 12
    run = _script(sr, n, groups, blocks)
    res = {}
    for traj in (True, False):
        p = _pool(ow, sr, n, traj)
        p.set_sample_rate(sr)
        p.stagger_tremolo(groups)
        assert p.tremolo_groups() == groups
        on, held, cap = p.trajectory_info()
        assert on == (n if traj else 0)
        res[traj] = run(p)
        if traj:                       # the reset engine and the warmed-up engine stand at their own t: two more phases
            assert p.tremolo_groups() == groups + 2
            d_t = [p[k].diag().tremolo_be_fallbacks for k in (0, 2, 4, 69)]
        else:
            d_g = [p[k].diag().tremolo_be_fallbacks for k in (0, 2, 4, 69)]
        p.close()
    for b in range(blocks):
        for what, i in (("R", 1), ("preamp", 2), ("out", 0)):
            assert np.array_equal(res[True][b][i], res[False][b][i]), (sr, b, what, np.max(np.abs(res[True][b][i] - res[False][b][i])))
    assert d_t == d_g or all(x >= 0 for x in d_t)
    # ... and the oracle's, for engines in different situations (plain, reset mid-run, warmed up alone, last)
    step = max(1, int(int((2 if sr < 88200.0 else 1) * sr / 5.6) / groups))
    cs = {}
    for k in (0, 2, 4, 69):
        c = oracle.OracleEngine(sr)
        c.set_sample_rate(sr)
        c.advance_tremolo((k % groups) * step)
        cs[k] = c
    ref = run(None, cs)
    for b in range(blocks):
        for k in cs:
            co = ref[b][k][0]
            rep = oracle.parity_report(res[True][b][0][k], co, abs_floor=oracle.ABS_FLOOR_OUTPUT)
            assert rep["n_bad"] == 0, (sr, b, k, rep)
    for c in cs.values():
        c.close()


def test_engines_older_than_the_store_continue_on_their_own_oscillator(hiplib, oracle):
    """OW_TREM_TRAJ_SECONDS = 0.9 s at 96 kHz chain rate: the warm-up alone (0.6 s) nearly fills the store, staggered engines cross
    its end one group after the other over the next blocks.  No seam: same bits as the per-group pool all the way."""
    import openwurli_amd as ow
    sr, n, groups = 48000.0, 12, 6
    hiplib.ow_test_clear_settle_caches()
    with _Env(OW_TREM_TRAJ_SECONDS="0.9"):
        pt = _pool(ow, sr, n, True)
    try:
        pg = _pool(ow, sr, n, False)
        outs = {}
        counts = []
        for name, p in (("t", pt), ("g", pg)):
            p.set_sample_rate(sr)
            p.stagger_tremolo(groups)
            for k in range(n):
                p[k].set_tremolo_depth(1.0); p[k].note_on(50 + k, 0.8)
            o = []
            for b in range(70):
                length = 512 if b % 5 else 211
                o.append((p.render(length).copy(), p.tremolo_r(2 * length).copy()))
                if name == "t":
                    counts.append(p.trajectory_info()[0])
            outs[name] = o
        cap = pt.trajectory_info()[2]
        assert cap == 90112                                        # 0.9 s x 96 kHz rounded up to a checkpoint boundary
        assert counts[0] == n and counts[-1] == 0 and sorted(counts, reverse=True) == counts and len(set(counts)) >= 4
        for b in range(70):
            assert np.array_equal(outs["t"][b][1], outs["g"][b][1]), (b, "R")
            assert np.array_equal(outs["t"][b][0], outs["g"][b][0]), (b, "out")
        # a reset brings an evicted engine back to t = 0 of the trajectory
        for p in (pt, pg):
            p[3].reset(); p[3].note_on(60, 0.9)
        assert pt.trajectory_info()[0] == 1
        for b in range(4):
            a, c = pt.render(256), pg.render(256)
            assert np.array_equal(a, c), b
            assert np.array_equal(pt.tremolo_r(512), pg.tremolo_r(512)), b
        pg.close()
    finally:
        pt.close()
        hiplib.ow_test_clear_settle_caches()                       # drop the tiny store: later tests get a full-size one


def test_pools_of_one_share_one_store(hiplib, oracle):
    """The plugin case: every instance is its own pool.  The first one extends the trajectory; an instance created later reads what
    is there (its t starts at 0) and both match their oracle engines."""
    import openwurli_amd as ow
    sr = 44100.0
    a = ow.WurliEngine(sr)
    a.set_sample_rate(sr)
    ca = oracle.OracleEngine(sr); ca.set_sample_rate(sr)
    for e in (a, ca):
        e.set_tremolo_depth(0.9); e.note_on(57, 0.8)
    for _ in range(20):
        rep = oracle.parity_report(a.render(441), ca.render(441), abs_floor=oracle.ABS_FLOOR_OUTPUT)
        assert rep["n_bad"] == 0, rep
    b = ow.WurliEngine(sr)
    b.set_sample_rate(sr)
    cb = oracle.OracleEngine(sr); cb.set_sample_rate(sr)
    for e in (b, cb):
        e.set_tremolo_depth(0.6); e.note_on(45, 0.9)
    for _ in range(10):
        for g, c in ((a, ca), (b, cb)):
            rep = oracle.parity_report(g.render(300), c.render(300), abs_floor=oracle.ABS_FLOOR_OUTPUT)
            assert rep["n_bad"] == 0, rep
    a.close(); b.close(); ca.close(); cb.close()


def test_prefetch_and_batch_tremolo_read_the_store(hiplib, oracle):
    """ow_tremolo_prefetch fills the store ahead of time; `preamp-bench render --tremolo-depth` jobs read it from t = 0 (Tremolo::new
    without a warm-up) and equal the per-call oscillator of OW_TREM_TRAJ=0 bit for bit."""
    import openwurli_amd as ow
    sr = 44100.0
    held = ow.tremolo_prefetch(sr, 1.5)
    assert held >= int(1.5 * 2 * sr)
    assert ow.tremolo_prefetch(sr, 0.1) >= held                    # never shrinks
    jobs = [dict(note=60, velocity=100, mlp=False, poweramp=False, volume=1.0, speaker=0.0, r_ldr=1e6, tremolo_depth=d) for d in (1.0, 0.4, 0.0)]
    with _Env(OW_TREM_TRAJ=None):
        x = ow.batch_render(jobs, sr, 1.0)
    with _Env(OW_TREM_TRAJ="0"):
        y = ow.batch_render(jobs, sr, 1.0)
    assert np.array_equal(x, y)
    assert not np.array_equal(x[0], x[2])


def test_latched_switches(hiplib):
    """The OW_* switches are read when a pool is created and never on the render path: flipping the environment under a live pool
    changes nothing, ow_test_pool_set_switch does."""
    import openwurli_amd as ow
    with _Env(OW_TREM_SERIAL="1", OW_PREAMP_WIDE="0"):
        p = ow.EnginePool(48000.0, 4)
    assert p.get_switch("trem_serial") == 1 and p.get_switch("preamp_wide") == 0
    os.environ["OW_PREAMP_WIDE"] = "1"
    try:
        p.render(64)
        assert p.get_switch("preamp_wide") == 0
    finally:
        del os.environ["OW_PREAMP_WIDE"]
    p.set_switch("preamp_wide", -1); p.set_switch("trem_serial", 0)
    assert p.get_switch("preamp_wide") == -1 and p.get_switch("trem_serial") == 0
    with pytest.raises(ow.OwError):
        p.set_switch("trem_traj", 0)
    p.close()


_FRESH = r"""
import json, sys, time
import numpy as np
import openwurli_amd as ow
sr, n = 48000.0, 256
p = ow.EnginePool(sr, n)              # no ow_tremolo_prefetch anywhere in this process
p.set_sample_rate(sr)                 # initialize(): chain build + 0.6 s warm-up, as the plugin does
p.stagger_tremolo(n)
for k in range(n):
    for note in range(33, 97):
        p[k].note_on(note, (40 + (37 * k) % 88) / 127.0)
time.sleep(0.5)                       # a host instantiates its plugins some time before the transport starts
rows = []
for b in range(40):
    st = p.trajectory_state()
    t0 = time.perf_counter()
    out = p.render(512)
    rows.append((st["complete"], st["oldest_t"], st["enqueued"], st["buffers"], st["capacity"], time.perf_counter() - t0))
print(json.dumps({"rows": rows, "finite": bool(np.all(np.isfinite(out))), "peak": float(np.max(np.abs(out)))}))
p.close()
"""


def test_fresh_process_never_waits_for_the_oscillator(hiplib):
    """A host that never calls ow_tremolo_prefetch (the reference has nothing to call: Tremolo::new settles inside the constructor,
    tremolo.rs:83-102): a FRESH process creates 256 instances with 256 tremolo phases and renders back to back (configs[4] literally).
    The store starts to run ahead when it is created (a helper thread feeds the oscillator whether or not anybody renders), so every
    sample a block needs is already complete when the block is asked for -- no block waits for the single oscillator -- and its buffers
    are the small first allocation (150 s), not the 1 800 s capacity.  (A pool that renders back to back outruns one oscillator --
    5.7 against 4 x real time -- so the lead shrinks while it does; 40 blocks stay far inside it.)"""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("OW_TREM_TRAJ_SECONDS", "OW_TREM_TRAJ_LEAD_SECONDS", "OW_TREM_TRAJ"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", _FRESH], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["finite"] and d["peak"] > 1e-3
    rows = d["rows"]
    n_os = 1024
    for b, (complete, oldest, enq, buffers, capacity, dt) in enumerate(rows):
        if b >= 3:
            assert complete >= oldest + n_os, (b, complete, oldest)          # everything the block reads was there before it was asked for
        assert enq <= buffers <= capacity
    up = lambda x: (x + 4095) // 4096 * 4096
    assert rows[0][3] == up(150 * 96000) and rows[0][4] == up(1800 * 96000)
    assert rows[0][2] - rows[0][1] > 64 * n_os                                # the feeder thread ran the oscillator ahead while the host was busy elsewhere


def test_store_grows_and_configures(hiplib, oracle):
    """ow_tremolo_configure + growth: a store limited to 3 s with a 0.5 s lead starts with buffers of ... 3 s (below the first allocation,
    nothing to grow); one limited to 400 s starts at 150 s and doubles when a reader comes within lead + 30 s of the end -- here forced
    by ow_tremolo_prefetch(200 s), which allocates where allocating is allowed.  Engines keep reading the same samples through the swap:
    the pool's output is bit-identical to a pool on per-group oscillators."""
    import openwurli_amd as ow
    sr = 44100.0                      # a chain rate no other test's store uses (88.2 kHz)
    ow.load_library().ow_test_clear_settle_caches()
    ow.tremolo_configure(400.0, 2.0)
    try:
        res = {}
        for traj in (True, False):
            p = _pool(ow, sr, 6, traj)
            p.set_sample_rate(sr)
            p.stagger_tremolo(3)
            for k in range(6):
                p[k].set_tremolo_depth(1.0); p[k].note_on(50 + k, 0.8)
            outs = []
            for b in range(8):
                if traj and b == 3:
                    st = p.trajectory_state()
                    assert st["buffers"] == 150 * 88200 // 4096 * 4096 + 4096 and st["capacity"] >= 400 * 88200
                    assert ow.tremolo_prefetch(sr, 200.0) >= 200 * 88200          # grows the buffers (150 s -> 200 s worth) and fills them
                    st = p.trajectory_state()
                    assert st["buffers"] >= 200 * 88200 and st["complete"] >= 200 * 88200
                outs.append(p.render(300).copy())
            res[traj] = outs
            p.close()
        for b in range(8):
            assert np.array_equal(res[True][b], res[False][b]), b
    finally:
        ow.tremolo_configure(0.0, -1.0)
        ow.load_library().ow_test_clear_settle_caches()


def _traj_kernels(hiplib, sr, n_settle, n, chunk, row, kick=None):
    import ctypes as C
    r = np.zeros(n, np.float64); st = np.zeros(18, np.float64)
    ck = np.zeros((n // 4096 + 2) * 16, np.float64); be = np.zeros(1024, np.uint64)
    ms = C.c_double(0.0)
    cold = np.zeros(2, np.uint64)
    rc = hiplib.ow_debug_trem_trajectory(C.c_double(sr), n_settle, n, chunk, row, r.ctypes.data, st.ctypes.data, ck.ctypes.data, be.ctypes.data,
                                         C.addressof(ms), 0, cold.ctypes.data, None if kick is None else np.ascontiguousarray(kick, np.float64).ctypes.data)
    assert rc == 0, hiplib.ow_last_error()
    _traj_kernels.cold = (int(cold[0]), int(cold[1]))      # generic sweeps / backward-Euler retries of this run (row kernels)
    return r, st, ck, be, ms.value


@pytest.mark.parametrize("sr", [48000.0, 44100.0, 96000.0])
def test_row_oscillator_kernels_equal_the_quad_lane_kernels(hiplib, sr):
    """ow_trem_row.h (one system per wavefront: lanes = matrix rows, zero coefficients for the emitted sparsity, the usual pivot order as a
    lane assignment, one branch per Newton sweep) against ow_trem_wide.h (the quad-lane step, itself bit-identical to the lane = engine
    step and to the oracle): from DC_OP through the growth of the oscillation into its settled regime (every sweep on the row step's fast
    path), and from states kicked off the operating point (the sweeps the row step hands to the generic sweep, the backward-Euler retry:
    below).  R, the state rows, every checkpoint and the fallback list must be the same bits, for any cut into launches."""
    n = 3 * 4096 + 1234
    a = _traj_kernels(hiplib, sr, 0, n, 4096, 0)
    b = _traj_kernels(hiplib, sr, 0, n, 4096, 1)
    c = _traj_kernels(hiplib, sr, 0, n, 1000, 1)
    for x in (b, c):
        assert np.array_equal(a[0].view(np.uint64), x[0].view(np.uint64)), int(np.argmax(a[0] != x[0]))
        assert np.array_equal(a[1].view(np.uint64), x[1].view(np.uint64)), (a[1], x[1])
        assert np.array_equal(a[2].view(np.uint64), x[2].view(np.uint64))
        assert np.array_equal(a[3], x[3])
    # settled regime (the regime the store extends in): 2 s of settle with either kernel, then one launch each
    n_settle = int(2 * (sr * 2 if sr < 88200.0 else sr))
    d = _traj_kernels(hiplib, sr, n_settle, 8192, 8192, 0)
    e = _traj_kernels(hiplib, sr, n_settle, 8192, 8192, 1)
    cold_settled = _traj_kernels.cold
    assert np.array_equal(d[0].view(np.uint64), e[0].view(np.uint64))
    assert np.array_equal(d[1].view(np.uint64), e[1].view(np.uint64))
    assert np.array_equal(d[2].view(np.uint64), e[2].view(np.uint64))
    assert d[0].min() > 10.0 and d[0].max() <= 1.0e6 and d[0].max() / d[0].min() > 3.0      # the cell swings (tremolo.rs:128-146)
    # From DC_OP the oscillation grows gently: no sweep ever needs the generic path (the counters say so).  A circuit KICKED off its
    # operating point does -- node voltages and junction currents moved by volts / milliamps: junction limiting, the 3.5 V cap, pivots off the
    # usual order, singular sweeps, backward-Euler retries -- and must still follow the quad-lane kernel bit for bit, through the recovery.
    worst = (0, 0)
    for k, kick in enumerate((np.array([0, 0, 2.0, 0, -1.5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0]),
                              np.array([-6.0, 3.0, 0, 4.0, 5.0, 0, 0.5, 1e-3, -1e-3, 2e-3, 1e-3, 0, 0, 0, 0, 0, 0, 0.0]),
                              np.array([9.0, -9.0, 9.0, -9.0, 9.0, 0, 1.0, 5e-3, 5e-3, -5e-3, 5e-3, -1e-3, 1e-3, 1e-3, -1e-3, 0, 0, 0.0]))):
        qa = _traj_kernels(hiplib, sr, 0, 6000, 4096, 0, kick=kick)
        ra = _traj_kernels(hiplib, sr, 0, 6000, 4096, 1, kick=kick)
        cold = _traj_kernels.cold
        worst = (max(worst[0], cold[0]), max(worst[1], cold[1]))
        assert np.array_equal(qa[0].view(np.uint64), ra[0].view(np.uint64)), (k, int(np.argmax(qa[0] != ra[0])), cold)
        assert np.array_equal(qa[1].view(np.uint64), ra[1].view(np.uint64)), (k, cold)
        assert np.array_equal(qa[2].view(np.uint64), ra[2].view(np.uint64)), (k, cold)
        assert np.array_equal(qa[3], ra[3]), (k, cold)
    assert worst[0] > 0, worst                                  # the kicks did hand sweeps to the generic sweep
    print(f"\n[oscillator step at {sr:.0f} Hz] quad-lane {d[4] * 1e3 / 8192:.3f} us, row {e[4] * 1e3 / 8192:.3f} us; generic sweeps / BE retries: "
          f"{cold_settled} in 8 192 settled steps, up to {worst} in 6 000 steps after a kick")


def test_trajectory_export_import_round_trip(hiplib, tmp_path):
    """ow_tremolo_export / ow_tremolo_import: the cross-process analogue of the reference's in-process start-up caches
    (dk_preamp/melange_adapter.rs:12-29 OnceLock; tremolo.rs:92-102).  A 256-engine pool on a GENERATED store and -- after every
    process-wide store and settled state has been dropped -- a fresh one on an IMPORTED store render the same bits (R rows and output);
    the import spares the new store both its settle and its oscillator steps; a file with a flipped sample, a truncated one and one for
    another chain rate are rejected, and so is a well-formed file whose middle was produced by something else when that reaches a
    checked segment (first / last)."""
    import openwurli_amd as ow
    sr, n, blocks, length = 48000.0, 256, 6, 512
    path = str(tmp_path / "traj_96k.bin")

    def run(p):
        p.stagger_tremolo(n)
        for k in range(0, n, 7):
            p[k].set_tremolo_depth(0.2 + 0.7 * (k % 5) / 5.0); p[k].note_on(40 + k % 50, 0.7)
        outs, rs = [], []
        for _ in range(blocks):
            outs.append(p.render(length).copy()); rs.append(p.tremolo_r(2 * length).copy())
        return outs, rs

    hiplib.ow_test_clear_settle_caches()
    a = ow.EnginePool(sr, n)
    ref = run(a)
    ow.tremolo_prefetch(sr, 3.0)
    wrote = ow.tremolo_export(sr, path)
    assert wrote >= 3 * 96000 // 4096 * 4096 and wrote % 4096 == 0
    a.close()
    size = os.path.getsize(path)
    assert size >= wrote * 8

    # rejected files leave no trace
    raw = bytearray(open(path, "rb").read())
    bad = bytearray(raw); bad[len(bad) // 2] ^= 1
    open(str(tmp_path / "flipped.bin"), "wb").write(bad)
    open(str(tmp_path / "short.bin"), "wb").write(raw[: len(raw) - 4096])
    hiplib.ow_test_clear_settle_caches()
    for name, why in (("flipped.bin", "checksum"), ("short.bin", "truncated")):
        with pytest.raises(ow.OwError, match=why):
            ow.tremolo_import(sr, str(tmp_path / name))
    with pytest.raises(ow.OwError, match="chain rate"):
        ow.tremolo_import(44100.0, path)

    # the import: no settle, no oscillator steps for what the file holds
    hiplib.ow_test_clear_settle_caches()
    took = ow.tremolo_import(sr, path)
    assert took == wrote
    assert ow.tremolo_import(sr, path) == 0                     # the store already holds it
    b = ow.EnginePool(sr, n)
    assert b.trajectory_info()[1] >= wrote
    got = run(b)
    for k in range(blocks):
        assert np.array_equal(ref[1][k].view(np.uint64), got[1][k].view(np.uint64)), (k, "R rows")
        assert np.array_equal(ref[0][k], got[0][k]), (k, "output")
    # ... and the store goes on from the file's end with its own oscillator: what it holds afterwards is what a store that never stopped holds
    assert ow.tremolo_prefetch(sr, 3.6) > wrote
    path2, path3 = str(tmp_path / "after_import.bin"), str(tmp_path / "generated.bin")
    n2 = ow.tremolo_export(sr, path2)
    b.close()
    hiplib.ow_test_clear_settle_caches()
    ow.tremolo_prefetch(sr, 3.6)
    n3 = ow.tremolo_export(sr, path3)

    def samples(pth, count):
        hdr = 240                                               # sizeof(TrajFileHeader)
        assert int(np.fromfile(pth, dtype=np.uint64, count=1, offset=40)[0]) >= count
        return np.fromfile(pth, dtype=np.uint64, count=count, offset=hdr)
    m = min(n2, n3)
    assert m > wrote
    assert np.array_equal(samples(path2, m), samples(path3, m))
    assert np.array_equal(samples(path, wrote), samples(path3, wrote))
