"""Tremolo phase groups: engines whose Twin-T / CdS state is bit-identical share one oscillator (tremolo.rs:121 -- the oscillator has no
audio input and no dependence on the depth knob, so R[n] is a function of the samples since the chain was built).  The sharing must
be invisible: an engine reset or warmed up on its own gets its own oscillator (per-engine fallback), also when it was the group's
leader; results stay the oracle's for every engine."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _groups(hiplib, pool):
    return hiplib.ow_test_pool_tremolo_groups(pool._h)


def test_individual_resets_split_the_group_and_match_the_oracle(hiplib, oracle):
    import openwurli_amd as ow
    sr, n = 48000.0, 6
    g = ow.EnginePool(sr, n)
    cs = [oracle.OracleEngine(sr) for _ in range(n)]
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)
    assert _groups(hiplib, g) == 1
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_tremolo_depth(1.0 - 0.1 * k); e.set_volume(0.5)
            e.note_on(45 + 6 * k, 0.85)

    def compare(tag, blocks, length=512):
        for b in range(blocks):
            go = g.render(length)
            gr = g.tremolo_r(2 * length)
            for k in range(n):
                co, _, _, cr = cs[k].render_taps(length)
                # engines at (nearly) full tremolo depth: the reference's own one-ulp indeterminacy reaches 1.9e-9 there
                # (tests/test_oracle_sensitivity.py::test_full_depth_tremolo_floor), the dense-play floor applies
                floor = oracle.ABS_FLOOR_DENSE if 1.0 - 0.1 * k >= 0.9 else oracle.ABS_FLOOR_OUTPUT
                rep = oracle.parity_report(go[k], co, abs_floor=floor)
                assert rep["n_bad"] == 0, (tag, b, k, rep["worst_ratio"], rep)
                d = 1.0 - 0.1 * k if tag != "start" or b > 0 else None     # depth ramps in the first block; the R check needs the final depth
                if d is not None and b > 0:
                    r = gr[k]
                    up, lo = 50000.0 * (1.0 - d), 50000.0 * d
                    top = up * 18000.0 / (up + 18000.0) if up > 0 else 0.0
                    shunt = top + (lo * (680.0 + r) / (lo + 680.0 + r) if lo > 0 else 0.0)
                    assert np.max(np.abs(shunt - cr) / cr) < 1e-9, (tag, b, k)
    compare("start", 3)
    g[2].reset(); cs[2].reset()                     # a member leaves: its own oscillator from the DC point, 2 s settle, 0.6 s warm-up
    assert _groups(hiplib, g) == 2
    for e in (g[2], cs[2]):
        e.note_on(70, 0.7)
    compare("member reset", 3)
    g[0].reset(); cs[0].reset()                     # the LEADER leaves: the rest of the group must keep the old oscillator
    assert _groups(hiplib, g) == 3
    for e in (g[0], cs[0]):
        e.note_on(52, 0.9)
    compare("leader reset", 3, 300)
    g[4].warm_up(); cs[4].warm_up()                 # 0.6 s of render() on one engine only: it runs ahead of its group
    assert _groups(hiplib, g) == 4
    compare("warm-up of one engine", 2)
    a = g.render(64)                                # engines 1, 3, 5 still share the pool's original oscillator
    r = g.tremolo_r(128)
    assert np.array_equal(r[1], r[3]) and np.array_equal(r[1], r[5])
    assert not np.array_equal(r[1], r[0]) and not np.array_equal(r[1], r[2]) and not np.array_equal(r[0], r[2]) and not np.array_equal(r[1], r[4])
    for c in cs:
        c.render(64)
    g.reset()                                       # whole-pool reset: one group again
    for c in cs:
        c.reset()
    assert _groups(hiplib, g) == 1
    for k in range(n):
        for e in (g[k], cs[k]):
            e.note_on(60 + k, 0.8)
    compare("pool reset", 3)
    g.close()


def test_staggered_groups_run_the_same_oscillator_at_different_times(hiplib):
    """ow_test_pool_stagger_tremolo: group k runs k * step samples ahead of group 0, so its R stream is group 0's stream shifted --
    bit for bit (the oscillator is autonomous), with both tremolo kernels, and members of one group stay bit-identical."""
    import openwurli_amd as ow
    sr, n, G = 48000.0, 100, 25
    step = max(1, int(int(2 * sr / 5.6) / G))
    streams = {}
    outs = {}
    for wide in ("0", "1"):
        os.environ["OW_TREM_WIDE"] = wide
        os.environ["OW_TREM_TRAJ"] = "0"            # per-group oscillators (the default pool reads the shared trajectory: test_gpu_trajectory.py)
        try:
            p = ow.EnginePool(sr, n)
            assert p.get_switch("trem_traj") == 0
            p.set_sample_rate(sr)
            assert hiplib.ow_test_pool_stagger_tremolo(p._h, G) == 0
            assert _groups(hiplib, p) == G
            for k in range(n):
                p[k].set_tremolo_depth(1.0); p[k].note_on(60, 0.8)
            rs, os_ = [], []
            for length in (512, 512, 100, 512, 512, 512, 512):
                os_.append(p.render(length))
                rs.append(p.tremolo_r(2 * length))
            p.close()
        finally:
            del os.environ["OW_TREM_WIDE"]
            del os.environ["OW_TREM_TRAJ"]
        streams[wide] = np.concatenate(rs, axis=1)
        outs[wide] = np.concatenate(os_, axis=1)
    assert np.array_equal(streams["0"], streams["1"]) and np.array_equal(outs["0"], outs["1"])
    r = streams["1"]
    for k in (1, 7, 24):
        sh = k * step
        assert np.array_equal(r[k, :r.shape[1] - sh], r[0, sh:]), k
        assert np.array_equal(r[k], r[k + G]) and np.array_equal(outs["1"][k], outs["1"][k + 3 * G])
    assert not np.array_equal(outs["1"][0], outs["1"][1])          # same note, different tremolo phase
