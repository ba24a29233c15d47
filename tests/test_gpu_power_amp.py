"""Melange 7-BJT power amp + rail dynamics on the GPU (power_amp.rs:65-165,279-465; gen_power_amp.rs) against the
oracle: the amp's own output per chain-rate sample (solver tap), the engine output, rail voltages, diag counters, set_rail_sag,
reset / set_sample_rate, and a FORCED divergence whose guard reset must land on the same sample on both sides."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PA = 1   # OW_POWER_AMP_MELANGE


def _pair(ow, oracle, sr, n=1, preamp_kind=0):
    g = ow.EnginePool(sr, n, preamp_kind=preamp_kind, power_amp_kind=PA)
    cs = [oracle.OracleEngine(sr, preamp_kind=preamp_kind, power_amp_kind=PA) for _ in range(n)]
    g.enable_power_amp_tap()
    return g, cs


def test_solver_alone_follows_the_oracle_bit_for_bit(hiplib, oracle):
    """The amp by itself on given input (ow_debug_power_amp against the oracle's adapter): sines from small signal to hard clipping, noise,
    at the codegen rate (88.2 kHz, baked matrices) and at 96 kHz (rebuilt), rail sag on and off, with a node poked past 100 V and a NaN
    mid-stream.  The device runs the reference's operation order with IEEE division / sqrt and the arithmetic-only fast_exp, so outputs,
    Newton iteration counts and guard resets must be IDENTICAL on every row, outputs and rail voltages bit-identical on the smooth rows
    and within 1e-12 where pnjlim's logarithm (glibc vs the device library, <= 1 ulp apart) takes part: clipping, white noise, and the
    restart after a guard reset (measured 1e-14)."""
    import ctypes as C
    L = oracle.lib()
    n = 3600
    rng = np.random.default_rng(5)
    for sr in (88200.0, 96000.0):
        t = np.arange(n) / sr
        rows = [0.001 * np.sin(2 * np.pi * 1000 * t), 0.05 * np.sin(2 * np.pi * 220 * t), 0.3 * np.sin(2 * np.pi * 110 * t),
                5.0 * np.sin(2 * np.pi * 100 * t), 0.02 * rng.standard_normal(n), 0.05 * np.sin(2 * np.pi * 330 * t), 0.05 * np.sin(2 * np.pi * 330 * t)]
        poke_at = np.array([-1, -1, -1, -1, -1, 1500, 2100], dtype=np.int64)
        poke_node = np.array([0, 0, 0, 0, 0, 8, 6], dtype=np.int32)
        poke_val = np.array([0, 0, 0, 0, 0, 1e6, float("nan")])
        x = np.ascontiguousarray(np.stack(rows))
        for sag in (1, 0):
            out = np.zeros_like(x); taps = np.zeros(x.shape + (3,))
            assert hiplib.ow_debug_power_amp(sr, x.ctypes.data_as(C.c_void_p), x.shape[0], n, sag, poke_at.ctypes.data_as(C.c_void_p),
                                             poke_node.ctypes.data_as(C.c_void_p), poke_val.ctypes.data_as(C.c_void_p),
                                             out.ctypes.data_as(C.c_void_p), taps.ctypes.data_as(C.c_void_p), 0) == 0
            for r in range(x.shape[0]):
                co = np.zeros(n); ct = np.zeros((n, 3))
                L.owo_mpa_run(C.c_double(sr), x[r].ctypes.data_as(C.c_void_p), C.c_size_t(n), sag, C.c_longlong(int(poke_at[r])), int(poke_node[r]),
                              C.c_double(poke_val[r]), co.ctypes.data_as(C.c_void_p), ct.ctypes.data_as(C.c_void_p))
                assert np.array_equal(taps[r, :, 1], ct[:, 1]), ("guard resets", sr, sag, r)          # the guard fires on the same samples
                assert np.array_equal(taps[r, :, 0], ct[:, 0]), ("newton iterations", sr, sag, r)
                assert np.max(np.abs(out[r] - co)) < 1e-12 and np.max(np.abs(taps[r, :, 2] - ct[:, 2])) < 1e-12, (sr, sag, r, np.max(np.abs(out[r] - co)))
                if r in (0, 1, 2):      # smooth signals never reach pnjlim's logarithm: every bit agrees
                    assert np.array_equal(out[r], co) and np.array_equal(taps[r, :, 2], ct[:, 2]), (sr, sag, r, np.max(np.abs(out[r] - co)))
            assert taps[5, -1, 1] >= 1 and taps[6, -1, 1] >= 1 and taps[0, -1, 1] == 0
            assert np.max(np.abs(out[3])) > 0.85                                                        # the clipping row does clip


def _compare(oracle, g, cs, blocks, length, tag, osr=2, floor=None):
    floor = oracle.ABS_FLOOR_OUTPUT if floor is None else floor
    worst = 0.0
    for b in range(blocks):
        go = g.render(length)
        gp = g.power_amp_out(osr * length)
        for k, c in enumerate(cs):
            co, cp = c.render_pa_tap(length, osr=osr)
            # the amp's normalised output (+-1).  Its INPUT is the preamp's output, which carries the legacy preamp's Newton-stop floor
            # (2e-9 V, oracle_binding.ABS_FLOOR_PREAMP) x 0.25 drive x the amp's gain 69/22: the tap cannot be tighter than that; the
            # solver by itself is compared bit for bit in test_solver_alone_follows_the_oracle_bit_for_bit
            rp = oracle.parity_report(gp[k], cp, abs_floor=oracle.ABS_FLOOR_PREAMP)
            ro = oracle.parity_report(go[k], co, abs_floor=floor)
            assert rp["n_bad"] == 0, (tag, "amp tap", b, k, rp)
            assert ro["n_bad"] == 0, (tag, "out", b, k, ro)
            worst = max(worst, ro["worst_ratio"])
    return worst


@pytest.mark.parametrize("sr", [44100.0, 48000.0, 96000.0])
def test_engine_with_melange_power_amp(hiplib, oracle, sr):
    """44.1 kHz host = the amp at its codegen rate (88.2 kHz chain: baked matrices); 48 kHz = rebuilt matrices at 96 kHz; 96 kHz host =
    no oversampling, amp at 96 kHz in the base-rate loop."""
    import openwurli_amd as ow
    g, cs = _pair(ow, oracle, sr, n=2)
    osr = 2 if sr < 88200.0 else 1
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)
    for k in range(2):
        for e in (g[k], cs[k]):
            e.set_volume(0.6); e.set_tremolo_depth(0.4 + 0.5 * k); e.set_speaker_character(0.3 * k)
            for n in ((45, 57, 64) if k == 0 else (40, 52, 60, 67, 76, 88)):
                e.note_on(n, 0.95)
    _compare(oracle, g, cs, 10, 256, ("chords", sr), osr=osr)
    for k in range(2):
        dg = g[k].power_amp_diag()
        dc = cs[k].power_amp_diag()
        assert (dg.clamp_count, dg.nr_max_iter_count, dg.guard_resets) == (dc[0], dc[1], dc[3])
        assert abs(dg.peak_output_volts - dc[2]) <= 1e-7      # volts at the amp's output node: the preamp's 2e-9 V Newton floor x 0.25 x 69
        assert g[k].rail_sag_enabled() and 22.5 < dg.rail_pos_volts <= 24.5
    # rail sag off on engine 0 (offsets zero from the next block on), then on again; engine 1 is re-struck meanwhile
    g[0].set_rail_sag(False); cs[0].set_rail_sag(False)
    for e in (g[1], cs[1]):
        e.note_off(60); e.note_on(60, 1.0)
    _compare(oracle, g, cs, 4, 256, ("sag off", sr), osr=osr)
    assert not g[0].rail_sag_enabled() and g[0].power_amp_diag().rail_pos_volts == 22.5
    g[0].set_rail_sag(True); cs[0].set_rail_sag(True)
    _compare(oracle, g, cs, 3, 333, ("sag on again", sr), osr=osr)
    g.close()


def test_forced_divergence_guard_lands_on_the_same_sample(hiplib, oracle):
    """power_amp.rs:373-421: a node past 100 V trips the guard -- state back to the settled point, rails reset, output = last good sample.
    Both sides are poked between two blocks; the hold, the reset and the recovery must coincide sample for sample."""
    import openwurli_amd as ow
    sr = 48000.0
    g, cs = _pair(ow, oracle, sr, n=3)
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)
    for k in range(3):
        for e in (g[k], cs[k]):
            e.set_volume(0.5); e.set_tremolo_depth(0.5)
            for n in (48, 55, 60, 64, 67):
                e.note_on(n, 0.9)
    _compare(oracle, g, cs, 4, 256, "before")
    # engine 1: NODE_OUT (8) forced to 1e6 V; engine 2: a NaN on the inverting input node (6): the solver's own NaN reset + the guard
    assert hiplib.ow_test_engine_poke_power_amp_node(g[1]._h, 8, 1e6) == 0
    cs[1].poke_power_amp_node(8, 1e6)
    assert hiplib.ow_test_engine_poke_power_amp_node(g[2]._h, 6, float("nan")) == 0
    cs[2].poke_power_amp_node(6, float("nan"))
    _compare(oracle, g, cs, 6, 256, "after the poke")
    d = [g[k].power_amp_diag() for k in range(3)]
    oc = [cs[k].power_amp_diag() for k in range(3)]
    # (the guard also fires on its own under chords -- the reference documents the solver's intermittent divergence under polyphonic input,
    # power_amp.rs:375-407 -- and it does so on the same samples on both sides, or the comparisons above would have failed)
    assert [x.guard_resets for x in d] == [x[3] for x in oc]
    assert d[1].guard_resets >= 1 and d[2].guard_resets >= 1
    assert [x.nr_max_iter_count for x in d] == [x[1] for x in oc]
    g.close()


def test_reset_and_pool_of_many(hiplib, oracle):
    """reset() = PowerAmp::reset (settled state, rails back to 22.5 V, last_good kept) inside WurliEngine::reset; a pool of 70 engines with
    different scripts (two wavefronts of the lane = engine kernel, ragged second one)."""
    import openwurli_amd as ow
    sr, n = 44100.0, 70
    g = ow.EnginePool(sr, n, power_amp_kind=PA)
    g.enable_power_amp_tap()
    picks = (0, 1, 63, 64, 69)
    cs = {k: oracle.OracleEngine(sr, power_amp_kind=PA) for k in picks}
    for k in range(n):
        g[k].set_tremolo_depth((k % 5) / 4.0)
        for nn in (40 + k % 30, 60 + k % 20):
            g[k].note_on(nn, 0.5 + 0.5 * ((k * 7) % 10) / 10.0)
    for k, c in cs.items():
        c.set_tremolo_depth((k % 5) / 4.0)
        for nn in (40 + k % 30, 60 + k % 20):
            c.note_on(nn, np.float32(0.5 + 0.5 * ((k * 7) % 10) / 10.0))
    for b in range(6):
        go = g.render(256)
        for k, c in cs.items():
            rep = oracle.parity_report(go[k], c.render(256), abs_floor=oracle.ABS_FLOOR_OUTPUT)
            assert rep["n_bad"] == 0, ("pool", b, k, rep)
    g[64].reset(); cs[64].reset()
    g[1].reset(); cs[1].reset()
    for k in (1, 64):
        for e in (g[k], cs[k]):
            e.note_on(72, 0.8)
    for b in range(4):
        go = g.render(300)
        for k, c in cs.items():
            rep = oracle.parity_report(go[k], c.render(300), abs_floor=oracle.ABS_FLOOR_OUTPUT)
            assert rep["n_bad"] == 0, ("after reset", b, k, rep)
    assert np.all(np.isfinite(go)) and np.max(np.abs(go)) > 1e-3
    g.close()


def test_behavioural_pool_reports_no_rails(hiplib):
    import openwurli_amd as ow
    e = ow.WurliEngine(48000.0)
    e.set_rail_sag(True)                               # a no-op: the behavioural amp has no separable rails (power_amp.rs:262-272)
    d = e.power_amp_diag()
    assert not e.rail_sag_enabled() and (d.clamp_count, d.nr_max_iter_count, d.peak_output_volts) == (0, 0, 0.0)
    e.close()


def test_amp_alone_on_a_dense_performance_shares_every_guard_event(hiplib, oracle):
    """The amp on IDENTICAL input, this time a realistic one: the preamp stream of the soak script (random dense play on four engines,
    0.75 s, volumes up to 0.65) as the oracle computes it, fed to the oracle's amp (inside its engine) and to ow_debug_power_amp.
    Under this script the divergence guard fires many times per engine; on equal input every one of those resets must land on the
    same sample, and the outputs must agree to the pnjlim-logarithm bound -- which is what the engine-level comparison cannot show
    once the two preamps differ by their floor (test_melange_power_amp_guard_timing_is_not_one_ulp_stable)."""
    import ctypes as C
    sr, length, n = 48000.0, 512, 4
    # fresh engines (WurliEngine::new, no warm-up) so that the amp starts from its settled clone exactly like ow_debug_power_amp;
    # two oracle engines per part in lock step: one hands out the preamp tap, the other the amp tap
    a = [oracle.OracleEngine(sr, power_amp_kind=PA) for _ in range(n)]
    b = [oracle.OracleEngine(sr, power_amp_kind=PA) for _ in range(n)]
    rng = np.random.default_rng(99)
    for k in range(n):
        for e in (a[k], b[k]):
            e.set_tremolo_depth(0.25 * k); e.set_volume(0.35 + 0.1 * k); e.set_speaker_character(0.3 * (k % 3))
    held = [[] for _ in range(n)]
    pre, amp = [[] for _ in range(n)], [[] for _ in range(n)]
    for _ in range(int(0.75 * sr / length)):
        for k in range(n):
            if rng.random() < 0.08 + 0.03 * k:
                note, vel = int(rng.integers(33, 97)), float(rng.uniform(0.2, 1.0))
                for e in (a[k], b[k]):
                    e.note_on(note, vel)
                held[k].append(note)
            if held[k] and rng.random() < 0.07:
                note = held[k].pop(int(rng.integers(0, len(held[k]))))
                for e in (a[k], b[k]):
                    e.note_off(note)
        for k in range(n):
            _, _, p, _ = a[k].render_taps(length)
            _, y = b[k].render_pa_tap(length)
            pre[k].append(p); amp[k].append(y)
    x = np.ascontiguousarray(np.stack([np.concatenate(p) for p in pre]) * 0.25)        # FIXED_CIRCUIT_DRIVE, engine.rs:544-546
    want = np.stack([np.concatenate(y) for y in amp])
    guards = [e.power_amp_diag()[3] for e in b]
    out = np.zeros_like(x); taps = np.zeros(x.shape + (3,))
    assert hiplib.ow_debug_power_amp(2.0 * sr, x.ctypes.data_as(C.c_void_p), x.shape[0], x.shape[1], 1, None, None, None,
                                     out.ctypes.data_as(C.c_void_p), taps.ctypes.data_as(C.c_void_p), 0) == 0
    print("guard resets per engine (oracle):", guards, " device:", taps[:, -1, 1].astype(int).tolist(),
          " max |out - oracle|: %.3e" % np.max(np.abs(out - want)))
    assert sum(guards) >= 4, guards                                                    # the script does exercise the guard
    assert taps[:, -1, 1].astype(int).tolist() == guards
    assert np.max(np.abs(out - want)) < 1e-11


def test_demand_ordered_dispatch_changes_no_sample(hiplib, oracle, monkeypatch):
    """k_post_mpa lets every engine of a wavefront walk its own sample counter (one Newton pass per trip) and, for blocks larger than
    the chip holds at once, dispatches the engines by falling demand of their last block (k_pa_order_*).  Both only change the
    schedule: a 300-engine pool with ten loudness classes rendered with the order forced on (OW_PA_SORT=2) and off (=0) is bit-identical,
    engine by engine, and the picked engines match the oracle."""
    import openwurli_amd as ow
    sr, n = 48000.0, 300
    picks = (0, 7, 8, 131, 299)

    def script(e, k):
        e.set_volume(0.5); e.set_tremolo_depth((k % 4) / 3.0)
        for nn in (36 + k % 24, 48 + k % 17, 60 + k % 13, 72 + k % 11):
            e.note_on(nn, np.float32(0.25 + 0.75 * ((k * 7) % 10) / 9.0))

    def run(mode):
        monkeypatch.setenv("OW_PA_SORT", mode)
        g = ow.EnginePool(sr, n, power_amp_kind=PA)
        for k in range(n):
            script(g[k], k)
        out = np.concatenate([g.render(256) for _ in range(5)], axis=1)
        d = [g[k].power_amp_diag() for k in picks]
        passes = g.power_amp_passes()                       # of the last block: 512 chain-rate samples, 1..70 passes each
        assert passes.shape == (n,) and passes.min() >= 512 and passes.max() <= 70 * 512
        g.close()
        return out, d, passes
    a, da, pa = run("2")
    b, db, pb = run("0")
    assert np.array_equal(pa, pb)                           # the same Newton work whichever wavefront an engine sat in
    assert np.array_equal(a, b)
    assert [(x.guard_resets, x.nr_max_iter_count) for x in da] == [(x.guard_resets, x.nr_max_iter_count) for x in db]
    assert len({a[k].tobytes() for k in range(40)}) > 20                # the engines do differ
    for k in picks:
        c = oracle.OracleEngine(sr, power_amp_kind=PA)
        script(c, k)
        co = np.concatenate([c.render(256) for _ in range(5)])
        rep = oracle.parity_report(a[k], co, abs_floor=oracle.ABS_FLOOR_OUTPUT)
        assert rep["n_bad"] == 0, (k, rep)
