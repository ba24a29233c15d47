"""The dispatch regime of the headline number, under the oracle.

`bench.py` measures one pool of 131 072 engines.  At that size the host side takes paths no small pool reaches: MIDI bursts cut into
per-thread slices, op packing by the worker threads (>= 4 096 dirty engines, openwurli_hip.hip render_range), voice lists packed in 32
slices, threaded post-render book-keeping (>= 16 384 engines), the lane-pair `k_preamp` / `k_post<true>` instead of the quad kernels
(> 4 096 engines), the lane = group `k_tremolo` (> 16 384 oscillators), and `render` straight into a pinned host block.  The wide
kernels are tied to these by bit-identity tests at small sizes; what was missing is the oracle next to the big pool itself.

Each run plays SURVEY 8d config 5 (instance k strikes all 64 keys at velocity (40 + 37k mod 88)/127, whole-keyboard note_off + note_on
at the 48 000-sample epoch, buffers of 512 cut sample-accurately at the epoch) for 1.1 s, compares engines {0, 1, 31, 32, 4095, 4096,
N/2, N-1} with one oracle engine each after EVERY block, and checks every engine of the pool against the engine 88 places before it
(same velocity, same phase group => the render must be bit-identical, whatever wavefront, slice or host thread handled it).
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 48000.0
EPOCH = 48000


def _velocity(k):
    return (40 + (37 * k) % 88) / 127.0


def _events(n_inst, restrike):
    from openwurli_amd import binding
    notes = np.arange(33, 97, dtype=np.uint8)
    vel = ((40 + (37 * np.arange(n_inst)) % 88) / 127.0).astype(np.float32)
    per = 64 * (2 if restrike else 1)
    ev = np.zeros((n_inst, per), dtype=np.dtype(binding.MIDI_DTYPE))
    ev["engine"] = np.arange(n_inst, dtype=np.uint32)[:, None]
    if restrike:
        ev["type"][:, 0::2] = 1; ev["note"][:, 0::2] = notes[None, :]
        ev["type"][:, 1::2] = 0; ev["note"][:, 1::2] = notes[None, :]; ev["value"][:, 1::2] = vel[:, None]
    else:
        ev["note"] = notes[None, :]; ev["value"] = vel[:, None]
    return ev.reshape(-1)


def _checked(n):
    return sorted({0, 1, 31, 32, 4095, 4096, n // 2, n - 1})


def _period(n, groups):
    """Engines k and k + period play the same velocity in the same tremolo phase group."""
    return 88 if groups in (None, 1) else int(np.lcm(88, groups))


def _run(hiplib, oracle, n_inst, total, modes, groups=None, restrike_at=EPOCH, trajectory=True):
    """modes: per block, cycled: 'host' (pageable numpy block), 'hbm' (left on the device, read back through the test hook),
    'pinned' (render straight into a page-locked block).  groups: ow_test_pool_stagger_tremolo(groups) after the warm-up."""
    import openwurli_amd as ow
    from openwurli_amd import binding
    import os
    saved = os.environ.get("OW_TREM_TRAJ")
    if not trajectory:
        os.environ["OW_TREM_TRAJ"] = "0"            # read when the pool is created: one oscillator per tremolo phase group (rounds 1-3)
    try:
        g = ow.EnginePool(SR, n_inst)
    finally:
        if not trajectory:
            if saved is None:
                del os.environ["OW_TREM_TRAJ"]
            else:
                os.environ["OW_TREM_TRAJ"] = saved
    assert g.get_switch("trem_traj") == (1 if trajectory else 0)
    g.set_sample_rate(SR)
    g.ensure_buffer_capacity(512)
    step = 0
    if groups:
        g.stagger_tremolo(groups)
        assert g.tremolo_groups() == groups
        step = max(1, int(int(2 * SR / 5.6) / groups))          # the hook's spacing: one oscillator period spread over the groups
    cs = {}
    for k in _checked(n_inst):
        c = oracle.OracleEngine(SR)
        c.set_sample_rate(SR)
        if groups:
            c.advance_tremolo((k % groups) * step)
        c.set_volume(0.5); c.set_tremolo_depth(0.5); c.set_speaker_character(0.0); c.set_mlp_enabled(True)
        cs[k] = c
    block = g.alloc_host_block(512) if "pinned" in modes else None
    period = _period(n_inst, groups)
    ev0, ev1 = _events(n_inst, False), _events(n_inst, True)
    pos, b, worst = 0, 0, 0.0
    scratch = np.zeros((n_inst, 512), dtype=np.float32)
    try:
        while pos < total:
            if pos == 0 or pos == restrike_at:
                g.midi(ev0 if pos == 0 else ev1)
                for k, c in cs.items():
                    for note in range(33, 97):
                        if pos:
                            c.note_off(note)
                        c.note_on(note, np.float32(_velocity(k)))
            length = min(512, total - pos)
            if pos < restrike_at:
                length = min(length, restrike_at - pos)
            mode = modes[b % len(modes)]
            if mode == "host":
                go = g.render(length)
            elif mode == "hbm":
                g.render(length, to_host=False)
                ptr, stride = g.device_output()
                assert stride == length
                go = scratch[:, :length] if length == 512 else np.zeros((n_inst, length), dtype=np.float32)
                go = np.ascontiguousarray(go)
                assert hiplib.ow_test_device_read(go.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), 4 * n_inst * length, 0) == 0
            else:
                g.render_into(block[0], length, length)                        # rows packed at the block length
                go = np.ctypeslib.as_array((C.c_float * (n_inst * length)).from_address(block[0])).reshape(n_inst, length).copy()
            for k, c in cs.items():
                rep = oracle.parity_report(go[k], c.render(length), abs_floor=oracle.ABS_FLOOR_OUTPUT)
                worst = max(worst, rep["worst_ratio"])
                assert rep["n_bad"] == 0, (n_inst, groups, mode, pos, k, rep)
            if groups is None or period < n_inst:
                assert np.array_equal(go[period:], go[:-period]), (n_inst, groups, mode, pos)
            assert np.all(np.isfinite(go))
            pos += length
            b += 1
        assert "voice dispatch" not in binding.last_error()
        for k, c in cs.items():
            assert [g[k].slot_state(i) for i in range(64)] == [c.slot_state(i) for i in range(64)], k
            d = g[k].diag()
            assert d.active_voices == c.active_voice_count() and d.nan_guard_fires == 0 and d.output_nan_resets == 0 and d.preamp_nan_resets == 0
            c.close()
        peak = float(np.max(np.abs(go)))
    finally:
        if block is not None:
            g.free_host_block(block)
        g.close()
    return worst, peak


TOTAL = int(1.1 * SR)


@pytest.mark.parametrize("modes", [("hbm",), ("pinned",)])
def test_pool_of_16384_against_oracle(hiplib, oracle, modes):
    worst, peak = _run(hiplib, oracle, 16384, TOTAL, modes)
    assert worst < 1.0 and 0.02 < peak < 4.0


def test_pool_of_131072_against_oracle(hiplib, oracle):
    """The bench's own pool size.  Blocks alternate between the three output paths."""
    worst, peak = _run(hiplib, oracle, 131072, TOTAL, ("hbm", "pinned", "host"))
    assert worst < 1.0 and 0.02 < peak < 4.0


def test_pool_of_16384_with_64_tremolo_groups(hiplib, oracle):
    """64 decorrelated phase groups (the quad-lane oscillator kernel with 64 leaders); engine k is compared with an oracle whose tremolo
    cell ran (k mod 64) * step samples ahead, engines 704 = lcm(88, 64) apart must agree bit for bit."""
    worst, peak = _run(hiplib, oracle, 16384, TOTAL, ("hbm", "pinned"), groups=64)
    assert worst < 1.0 and 0.02 < peak < 4.0


def test_pool_of_131072_fully_decorrelated(hiplib, oracle):
    """One tremolo phase per engine at the bench's pool size -- the bench's own configuration: 131 072 engines at 131 072 different t of
    the shared trajectory.  0.35 s with a whole-keyboard re-strike at 0.25 s."""
    worst, peak = _run(hiplib, oracle, 131072, int(0.35 * SR), ("hbm", "pinned"), groups=131072, restrike_at=12000)
    assert worst < 1.0 and 0.02 < peak < 4.0


def test_pool_of_32768_fully_decorrelated_lane_per_group_kernel(hiplib, oracle):
    """Smallest pool that takes the lane = group tremolo kernel with one group per engine, for the whole 1.1 s script.  Created under
    OW_TREM_TRAJ=0: since round 4 a default pool reads the shared trajectory and runs no oscillator of its own; this keeps the per-group
    path (what engines older than the trajectory store fall back to) under the oracle at pool scale."""
    worst, peak = _run(hiplib, oracle, 32768, TOTAL, ("hbm",), groups=32768, trajectory=False)
    assert worst < 1.0 and 0.02 < peak < 4.0


def test_pool_of_32768_fully_decorrelated_on_the_trajectory(hiplib, oracle):
    """The same pool and script on the shared trajectory (the default): 32 768 engines at 32 768 different t."""
    worst, peak = _run(hiplib, oracle, 32768, TOTAL, ("hbm",), groups=32768)
    assert worst < 1.0 and 0.02 < peak < 4.0


def test_melange_power_amp_pool_larger_than_the_chip_holds(hiplib, oracle):
    """`--power-amp melange` at the size its bench line runs at takes the demand-ordered dispatch of k_post_mpa on its own (more
    engines than the 16 384 the chip holds at once: DESIGN 14): a 17 408-engine pool, config-5 velocities on a 12-key chord, three
    blocks of 128 (the first orders by the zero demand of a fresh pool, the later ones by the passes of the block before), sampled
    engines against one oracle engine each, every engine against the engine 88 places before it (same velocity => bit-identical,
    wherever the order put it)."""
    import openwurli_amd as ow
    n = 17408
    g = ow.EnginePool(SR, n, power_amp_kind=1)
    keys = (36, 40, 43, 48, 52, 55, 60, 64, 67, 72, 76, 79)
    from openwurli_amd import binding
    vel = ((40 + (37 * np.arange(n)) % 88) / 127.0).astype(np.float32)
    ev = np.zeros((n, len(keys)), dtype=np.dtype(binding.MIDI_DTYPE))
    ev["engine"] = np.arange(n, dtype=np.uint32)[:, None]
    ev["note"] = np.array(keys, dtype=np.uint8)[None, :]
    ev["value"] = vel[:, None]
    g.midi(ev.reshape(-1))
    picks = sorted({0, 1, 87, 88, 4097, n // 2, n - 1})
    cs = {}
    for k in picks:
        c = oracle.OracleEngine(SR, power_amp_kind=1)
        for key in keys:
            c.note_on(key, np.float32(_velocity(k)))
        cs[k] = c
    try:
        for b in range(3):
            out = g.render(128)
            assert np.array_equal(out[88:], out[:-88]), ("engine k vs k - 88", b)
            assert len({out[k].tobytes() for k in range(88)}) > 60
            for k, c in cs.items():
                rep = oracle.parity_report(out[k], c.render(128), abs_floor=oracle.ABS_FLOOR_OUTPUT)
                assert rep["n_bad"] == 0, ("melange power amp, ordered dispatch", b, k, rep)
    finally:
        g.close()
