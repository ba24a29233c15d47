"""Host logic without a GPU: the voice-pool / MIDI state machine of the product library (detached engine hooks) against
the oracle's WurliEngine on random event sequences.  The oracle renders real audio (small blocks) so that voices are
freed by the genuine -80 dB / 10 s rules; the freed slots are fed to the product's bookkeeping as the "silent" mask the
GPU would report.  Compared after every block: every slot's state and note, steal-voice population, op stream."""
import ctypes as C

import numpy as np
import pytest


class Detached:
    def __init__(self, lib, sr):
        self.lib = lib
        self.h = C.c_void_p(lib.ow_test_engine_new(float(sr)))

    def close(self):
        self.lib.ow_test_engine_free(self.h)

    def note_on(self, n, v): self.lib.ow_engine_note_on(self.h, int(n) & 0xFF, float(v))
    def note_off(self, n): self.lib.ow_engine_note_off(self.h, int(n) & 0xFF)
    def set_sustain(self, h): self.lib.ow_engine_set_sustain(self.h, 1 if h else 0)
    def state(self, s): return self.lib.ow_engine_slot_state(self.h, s)
    def note(self, s): return self.lib.ow_engine_slot_note(self.h, s)
    def masks(self): return self.lib.ow_test_engine_masks(self.h, 0), self.lib.ow_test_engine_masks(self.h, 1)

    def take_ops(self):
        cap = 4096
        t = (C.c_uint8 * cap)(); sl = (C.c_uint8 * cap)(); nt = (C.c_uint8 * cap)(); sd = (C.c_uint32 * cap)(); ve = (C.c_double * cap)()
        n = self.lib.ow_test_engine_take_ops(self.h, t, sl, nt, sd, ve, cap)
        return [(t[i], sl[i], nt[i], sd[i], ve[i]) for i in range(n)]

    def after_render(self, length, silent_mask): self.lib.ow_test_engine_after_render(self.h, int(length), C.c_uint64(silent_mask))


def _compare_slots(d, o):
    for s in range(64):
        assert d.state(s) == o.slot_state(s), s
        if d.state(s) != 0:
            assert d.note(s) == o.slot_note(s), s


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_state_machine_matches_oracle(hiplib, oracle, seed):
    sr = 44100.0
    rng = np.random.default_rng(seed)
    d = Detached(hiplib, sr)
    o = oracle.OracleEngine(sr)
    sustain = False
    age = 0
    for step in range(160):
        # a burst of events, dense enough to overflow 64 voices regularly
        for _ in range(int(rng.integers(0, 14))):
            r = rng.random()
            note = int(rng.integers(30, 100))              # includes out-of-range notes (clamped to 33..96, engine.rs:300)
            if r < 0.62:
                vel = float(np.float32(rng.random()))
                d.note_on(note, vel); o.note_on(note, vel)
                age += 1
            elif r < 0.92:
                d.note_off(note); o.note_off(note)
            else:
                sustain = not sustain
                d.set_sustain(sustain); o.set_sustain(sustain)
        _compare_slots(d, o)
        ops = d.take_ops()
        for (t, sl, nt, sd, ve) in ops:
            assert t in (1, 2, 3) and sl < 64 and 33 <= nt <= 96
            if t == 3:
                assert sd == int(sr * 0.005)                # 5 ms steal crossfade (engine.rs:318)
        main_mask, steal_mask = d.masks()
        assert bin(steal_mask).count("1") == o.steal_voice_count()
        assert bin(main_mask).count("1") == o.active_voice_count()
        # render: the oracle decides which voices fell silent
        before = [o.slot_state(s) for s in range(64)]
        length = int(rng.choice([64, 200, 256, 1024, 4096]))
        o.render(length)
        silent = 0
        for s in range(64):
            if before[s] != 0 and o.slot_state(s) == 0:
                silent |= 1 << s
        d.after_render(length, silent)
        _compare_slots(d, o)
        assert bin(d.masks()[1]).count("1") == o.steal_voice_count()
    d.close()
    o.close()


def test_note_seeds_and_op_order(hiplib):
    """engine.rs:325-327: seed = note * 2654435761 + age_counter (u32 wrapping); a steal emits move-to-steal before the note-on."""
    d = Detached(hiplib, 48000.0)
    for n in range(33, 97):
        d.note_on(n, 0.5)
    ops = d.take_ops()
    assert [o[0] for o in ops] == [1] * 64
    for k, (t, sl, nt, sd, ve) in enumerate(ops):
        assert sl == k and nt == 33 + k and sd == ((33 + k) * 2654435761 + (k + 1)) & 0xFFFFFFFF and ve == 0.5
    d.note_on(60, 0.9)                                       # pool full: oldest Held (slot 0) is stolen
    ops = d.take_ops()
    assert [(o[0], o[1]) for o in ops] == [(3, 0), (1, 0)]
    assert ops[0][3] == 240 and ops[1][3] == (60 * 2654435761 + 65) & 0xFFFFFFFF
    d.note_off(60)                                           # oldest Held voice playing note 60 (slot 27), not the newest
    ops = d.take_ops()
    assert [(o[0], o[1]) for o in ops] == [(2, 27)]
    d.close()
