"""BASELINE configs[3] and configs[4] AT SIZE on the GPU (VERDICT r01 "configs untested").

configs[4]: ONE pool of 256 independent 64-voice engines (instance k plays velocity (40 + 37k mod 88)/127),
all 64 keys, 1.0 s re-strike epoch, buffers of 512, 1.2 s so that one whole-keyboard re-strike (release-steal + 5 ms crossfade of
all 64 slots) is inside; engines {0, 1, 31, 32, 63, 64, 128, 255} are compared with one oracle engine each after EVERY block
(the first / last lanes of the 32-engine preamp wavefronts and of the 64-engine tremolo wavefronts), determinism and diag are
checked over all 256.  Run once per OW_TREM_WIDE setting: a 256-engine pool takes the quad-lane tremolo kernel by default and
the lane = engine kernel when forced.

configs[3]: the full job grid of ml/render_model_notes.py:26,106-114 (64 notes x 8 velocity buckets = 512 jobs x 5 s) through
ONE ow_batch_render call; the 16 rows SURVEY 8d names are compared with the oracle *out of that call*, every row is checked for
finiteness / level / determinism, with both OW_CHAIN_WIDE settings (512 jobs take k_job_chain_wide by default).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 48000.0
CHECKED = (0, 1, 31, 32, 63, 64, 128, 255)


def _velocity(k):
    return (40 + (37 * k) % 88) / 127.0


class _Env:
    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = os.environ.get(self.name)
        if self.value is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.value

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.old


def _config5_events(n_inst, restrike):
    from openwurli_amd import binding
    notes = np.arange(33, 97, dtype=np.uint8)
    vel = np.array([_velocity(k) for k in range(n_inst)], dtype=np.float32)
    per = 64 * (2 if restrike else 1)
    ev = np.zeros((n_inst, per), dtype=np.dtype(binding.MIDI_DTYPE))
    ev["engine"] = np.arange(n_inst, dtype=np.uint32)[:, None]
    if restrike:
        ev["type"][:, 0::2] = 1; ev["note"][:, 0::2] = notes[None, :]
        ev["type"][:, 1::2] = 0; ev["note"][:, 1::2] = notes[None, :]; ev["value"][:, 1::2] = vel[:, None]
    else:
        ev["note"] = notes[None, :]; ev["value"] = vel[:, None]
    return ev.reshape(-1)


def _run_config4(ow, oracle, trem_wide, with_oracle):
    n_inst, total = 256, int(1.2 * SR)
    # trem_wide None = the default pool: shared tremolo trajectory + fused preamp / output launch; "0" / "1" = created under OW_TREM_TRAJ=0 with
    # the lane-per-group / quad-per-group oscillator kernel forced (the per-group path of rounds 1-3, bit-identical to the trajectory)
    with _Env("OW_TREM_WIDE", trem_wide), _Env("OW_TREM_TRAJ", None if trem_wide is None else "0"):
        g = ow.EnginePool(SR, n_inst)
        assert g.get_switch("trem_traj") == (1 if trem_wide is None else 0)
        g.set_sample_rate(SR)
        g.ensure_buffer_capacity(512)
        cs = {}
        if with_oracle:
            for k in CHECKED:
                c = oracle.OracleEngine(SR)
                c.set_sample_rate(SR)
                cs[k] = c
        for k in range(n_inst):      # the engine defaults, set explicitly like the plugin's sync_params does every block
            g[k].set_volume(0.5); g[k].set_tremolo_depth(0.5); g[k].set_speaker_character(0.0); g[k].set_mlp_enabled(True)
        for c in cs.values():
            c.set_volume(0.5); c.set_tremolo_depth(0.5); c.set_speaker_character(0.0); c.set_mlp_enabled(True)
        ev0, ev1 = _config5_events(n_inst, False), _config5_events(n_inst, True)
        pos, worst, outs = 0, 0.0, []
        while pos < total:
            if pos % 48000 == 0:
                g.midi(ev0 if pos == 0 else ev1)
                for k, c in cs.items():
                    for n in range(33, 97):
                        if pos:
                            c.note_off(n)
                        c.note_on(n, np.float32(_velocity(k)))
            length = min(512, 48000 - pos % 48000, total - pos)
            go = g.render(length)
            outs.append(go)
            for k, c in cs.items():
                rep = oracle.parity_report(go[k], c.render(length), abs_floor=oracle.ABS_FLOOR_OUTPUT)
                worst = max(worst, rep["worst_ratio"])
                assert rep["n_bad"] == 0, ("config4", trem_wide, pos, k, rep)
                assert g[k].active_voice_count() == c.active_voice_count(), (pos, k)
            pos += length
        diags = [g[k].diag() for k in range(n_inst)]
        for k, c in cs.items():
            assert [g[k].slot_state(i) for i in range(64)] == [c.slot_state(i) for i in range(64)], k
            c.close()
        from openwurli_amd import binding
        assert "voice dispatch" not in binding.last_error()
        g.close()
    return np.concatenate(outs, axis=1), diags, worst


@pytest.mark.parametrize("trem_wide", [None, "0"])
def test_config4_256_engine_pool_against_oracle(hiplib, oracle, trem_wide):
    import openwurli_amd as ow
    a, da, worst = _run_config4(ow, oracle, trem_wide, True)
    assert a.shape == (256, int(1.2 * SR)) and worst < 1.0
    assert np.all(np.isfinite(a)) and 0.02 < np.max(np.abs(a)) < 4.0
    assert all(d.active_voices == 64 and d.steal_voices == 0 and d.nan_guard_fires == 0 and d.preamp_nan_resets == 0
               and d.output_nan_resets == 0 and d.tremolo_be_fallbacks == 0 for d in da)
    # every instance sounds (peak within 30 dB of the loudest) and instances with different velocities differ
    peaks = np.max(np.abs(a), axis=1)
    assert peaks.min() > peaks.max() * 10 ** (-30 / 20)
    assert not np.array_equal(a[0], a[1])
    # instances 0 and 88 play the same velocity ((37 * 88) mod 88 == 0): bit-identical renders from different wavefronts
    assert np.array_equal(a[0], a[88]) and np.array_equal(a[1], a[89])
    # determinism over all 256 (second pool, same script, no oracle)
    b, db, _ = _run_config4(ow, oracle, trem_wide, False)
    assert np.array_equal(a, b)


def test_config4_tremolo_kernels_agree_at_256(hiplib, oracle):
    """The two tremolo kernels give the bit-identical pool render at this size (the R stream is bit-identical by construction)."""
    import openwurli_amd as ow
    a, _, _ = _run_config4(ow, oracle, "1", False)
    b, _, _ = _run_config4(ow, oracle, "0", False)
    assert np.array_equal(a, b)
    c, _, _ = _run_config4(ow, oracle, None, False)          # ... and equal the default pool's (trajectory, fused chain launch)
    assert np.array_equal(a, c)


SUBSET = [(n, v) for n in (33, 48, 60, 72, 84, 91, 96, 40) for v in (50, 127)]     # SURVEY 8d parity subset of config 4


def test_config3_full_batch_in_one_call(hiplib, oracle):
    import openwurli_amd as ow
    from openwurli_amd.distributed import model_notes_job_list
    jobs = model_notes_job_list()
    assert len(jobs) == 512
    dur = 5.0
    n = int(dur * SR)
    index = {(j["note"], j["velocity"]): i for i, j in enumerate(jobs)}
    peaks = {}
    ref = None
    for wide in (None, "0"):
        with _Env("OW_CHAIN_WIDE", wide):
            g = ow.batch_render(jobs, sample_rate=SR, duration_s=dur)
        assert g.shape == (512, n)
        assert np.all(np.isfinite(g))
        pk = np.max(np.abs(g), axis=1)
        assert pk.min() > 1e-4 and pk.max() < 8.0, (wide, pk.min(), pk.max())
        # louder velocity bucket -> larger peak, for every note (render_model_notes buckets are ascending)
        grid = pk.reshape(64, 8)
        assert np.all(grid[:, -1] > grid[:, 0])
        for (note, vel) in SUBSET:                      # oracle on the 16 rows OF THIS CALL
            i = index[(note, vel)] if (note, vel) in index else None
            if i is None:                               # velocity 50 is a bucket, 127 is a bucket; both are in the grid
                raise AssertionError((note, vel))
            c = oracle.batch_render_job(note, vel, dur, SR)
            rep = oracle.parity_report(g[i], c, abs_floor=oracle.ABS_FLOOR_BATCH)
            assert rep["n_bad"] == 0, ("config3", wide, note, vel, rep)
        if ref is None:
            ref = g
        else:
            assert np.array_equal(ref, g)               # k_job_chain_wide == k_job_chain bit for bit, at the full grid
        peaks[wide] = pk
        del g
    # determinism of one setting
    with _Env("OW_CHAIN_WIDE", None):
        again = ow.batch_render(jobs, sample_rate=SR, duration_s=dur)
    assert np.array_equal(ref, again)
