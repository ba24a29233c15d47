"""Host mirror of the reference's engine API over the C-ABI.

Method names, argument meaning and error behaviour follow ``WurliEngine``
(/root/reference/crates/openwurli-dsp/src/engine.rs:194-675) so that parity tests read like
the reference's own engine tests.  Nothing here computes audio: every call forwards to
``libopenwurli_hip.so``.
"""
import ctypes as C
import enum

import numpy as np

from . import binding
from .binding import OwDiag, OwError, OwPowerAmpDiag


class VoiceState(enum.IntEnum):  # engine.rs:30-37
    Free = 0
    Held = 1
    Sustained = 2
    Releasing = 3


class _EngineHandle:
    """Methods shared by a pool-of-one ``WurliEngine`` and the engines of an ``EnginePool``."""

    def __init__(self, lib, handle):
        self._lib = lib
        self._h = C.c_void_p(handle)

    # ---- MIDI (engine.rs:299-374)
    def note_on(self, note, velocity):
        self._lib.ow_engine_note_on(self._h, int(note) & 0xFF, float(velocity))

    def note_off(self, note):
        self._lib.ow_engine_note_off(self._h, int(note) & 0xFF)

    def set_sustain(self, held):
        self._lib.ow_engine_set_sustain(self._h, 1 if held else 0)

    # ---- params (engine.rs:378-400)
    def set_volume(self, v):
        self._lib.ow_engine_set_volume(self._h, float(v))

    def set_tremolo_depth(self, d):
        self._lib.ow_engine_set_tremolo_depth(self._h, float(d))

    def set_speaker_character(self, c):
        self._lib.ow_engine_set_speaker_character(self._h, float(c))

    def set_mlp_enabled(self, on):
        self._lib.ow_engine_set_mlp_enabled(self._h, 1 if on else 0)

    def set_noise_enabled(self, on):
        self._lib.ow_engine_set_noise_enabled(self._h, 1 if on else 0)

    def set_noise_gain(self, g):
        self._lib.ow_engine_set_noise_gain(self._h, float(g))

    def set_rail_sag(self, on):
        """engine.rs:406-408 (melange power amp; a no-op on the behavioural one)."""
        self._lib.ow_engine_set_rail_sag(self._h, 1 if on else 0)

    def rail_sag_enabled(self):
        return bool(self._lib.ow_engine_rail_sag_enabled(self._h))

    def power_amp_diag(self):
        """engine.rs:418-420 ``(clamp_count, nr_max_iter_count, peak_output_volts)`` as the first three fields of the returned struct."""
        d = OwPowerAmpDiag()
        self._lib.ow_engine_power_amp_diag(self._h, C.byref(d))
        return d

    def set_noise_seed(self, seed):
        """gen_preamp::set_seed of the melange preamp's main state (0 = process-wide clock entropy, the reference's only mode)."""
        self._lib.ow_engine_set_noise_seed(self._h, int(seed))

    def poke_voice(self, slot, steal, field, value):
        """Test hook (openwurli_hip_test.h): overwrite one double of a voice record on the device."""
        return self._lib.ow_test_engine_poke_voice(self._h, int(slot), 1 if steal else 0, int(field), float(value))

    def poke_preamp_node(self, node, volts, shadow=False):
        """Test hook (openwurli_hip_test.h): overwrite a node voltage of the legacy preamp's main / shadow solver state."""
        return self._lib.ow_test_engine_poke_preamp_node(self._h, 1 if shadow else 0, int(node), float(volts))

    def read_preamp_state(self, shadow=False):
        """Test hook (openwurli_hip_test.h): the legacy preamp's solver state after the blocks rendered so far -- j_cin, cin_rhs_prev, v[8],
        i_nl[2], v_nl[2] (DkState, dk_preamp_legacy.rs:231-239)."""
        import numpy as np
        out = np.zeros(14, dtype=np.float64)
        if self._lib.ow_test_engine_read_preamp_state(self._h, 1 if shadow else 0, out.ctypes.data) != 0:
            raise RuntimeError("ow_test_engine_read_preamp_state failed")
        return out

    def reset(self):
        self._lib.ow_engine_reset(self._h)
        binding.raise_if_error(self._lib)

    def warm_up(self):
        self._lib.ow_engine_warm_up(self._h)
        binding.raise_if_error(self._lib)

    # ---- introspection (engine.rs:606-675)
    def diag(self):
        d = OwDiag()
        self._lib.ow_engine_get_diag(self._h, C.byref(d))
        return d

    def active_voice_count(self):
        return self.diag().active_voices

    def held_voice_count(self):
        return self.diag().held_voices

    def sustained_voice_count(self):
        return self.diag().sustained_voices

    def slot_state(self, slot):
        """VoiceSlot.state of one of the 64 slots (VoiceState value)."""
        return self._lib.ow_engine_slot_state(self._h, int(slot))

    def slot_note(self, slot):
        """VoiceSlot.midi_note of one of the 64 slots."""
        return self._lib.ow_engine_slot_note(self._h, int(slot))

    def count_voices_in_state(self, state):
        return sum(1 for s in range(64) if self._lib.ow_engine_slot_state(self._h, s) == int(state))

    def count_voices_with_note_in_state(self, note, state):
        return sum(1 for s in range(64)
                   if self._lib.ow_engine_slot_state(self._h, s) == int(state)
                   and self._lib.ow_engine_slot_note(self._h, s) == int(note))

    def has_steal_voice_for(self, note):
        return bool(self._lib.ow_engine_has_steal_voice_for(self._h, int(note) & 0xFF))

    def nan_guard_fires(self):
        return self.diag().nan_guard_fires

    def is_sustain_held(self):
        return bool(self.diag().sustain_held)


class WurliEngine(_EngineHandle):
    """One engine on one GPU (a pool of one).  ``WurliEngine(sr)`` == ``WurliEngine::new(sr)``."""

    def __init__(self, sample_rate, device=0, preamp_kind=0, power_amp_kind=0, tremolo_kind=0):
        lib = binding.load_library()
        h = lib.ow_engine_new_kinds(float(sample_rate), int(device), int(preamp_kind), int(power_amp_kind), int(tremolo_kind))
        if not h:
            raise OwError(binding.take_error(lib))
        super().__init__(lib, h)

    def close(self):
        if self._h:
            self._lib.ow_engine_free(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_sample_rate(self, sr):
        self._lib.ow_engine_set_sample_rate(self._h, float(sr))
        binding.raise_if_error(self._lib)

    def ensure_buffer_capacity(self, n):
        self._lib.ow_engine_ensure_buffer_capacity(self._h, int(n))
        binding.raise_if_error(self._lib)

    def render(self, out):
        """``engine.render(&mut out)``: ``out`` is a writable 1-D float32 numpy array, or a length."""
        if isinstance(out, (int, np.integer)):
            out = np.zeros(int(out), dtype=np.float32)
        assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"]
        self._lib.ow_engine_render(self._h, out.ctypes.data_as(C.c_void_p), out.size)
        binding.raise_if_error(self._lib)          # `out` holds silence in that case (the C contract)
        return out


class EnginePool:
    """I independent engines rendered in lock-step (lane = engine on the GPU)."""

    def __init__(self, sample_rate, n_engines, device=0, preamp_kind=0, power_amp_kind=0, tremolo_kind=0):
        self._lib = binding.load_library()
        h = self._lib.ow_pool_new_kinds(float(sample_rate), int(n_engines), int(device), int(preamp_kind), int(power_amp_kind), int(tremolo_kind))
        if not h:
            raise OwError(binding.take_error(self._lib))
        self._h = C.c_void_p(h)
        self.n = int(n_engines)
        self.device = int(device)
        self._engines = {}

    def close(self):
        if self._h:
            self._lib.ow_pool_free(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getitem__(self, i):
        i = int(i)
        if not 0 <= i < self.n:
            raise IndexError(i)
        if i not in self._engines:
            self._engines[i] = _EngineHandle(self._lib, self._lib.ow_pool_engine(self._h, i))
        return self._engines[i]

    def set_sample_rate(self, sr):
        if self._lib.ow_pool_set_sample_rate(self._h, float(sr)) != 0:
            raise OwError(binding.take_error(self._lib))

    def reset(self):
        self._lib.ow_pool_reset(self._h)
        binding.raise_if_error(self._lib)

    def ensure_buffer_capacity(self, n):
        self._lib.ow_pool_ensure_buffer_capacity(self._h, int(n))
        binding.raise_if_error(self._lib)

    def render(self, length, to_host=True):
        """Render ``length`` samples on every engine; returns float32 [n, length] (or None if left in HBM)."""
        if to_host:
            out = np.zeros((self.n, int(length)), dtype=np.float32)
            self._lib.ow_pool_render(self._h, out.ctypes.data_as(C.c_void_p), int(length), int(length))
            binding.raise_if_error(self._lib)
            return out
        self._lib.ow_pool_render(self._h, None, 0, int(length))
        binding.raise_if_error(self._lib)
        return None

    def render_into(self, host_ptr, stride, length):
        """``ow_pool_render`` into caller memory: float32 [n, stride] at address ``host_ptr`` (e.g. a block from ``alloc_host_block``)."""
        self._lib.ow_pool_render(self._h, C.c_void_p(int(host_ptr)), int(stride), int(length))
        binding.raise_if_error(self._lib)

    def alloc_host_block(self, length, device=0):
        """Page-locked float32 [n, length] block for ``render_into`` -> (address, stride)."""
        ptr = self._lib.ow_host_alloc(4 * self.n * int(length), int(device))
        if not ptr:
            raise OwError(binding.take_error(self._lib))
        return (ptr, int(length))

    def alloc_host_events(self, n_events, device=0):
        """Page-locked event list for ``midi``: a numpy structured array (dtype binding.MIDI_DTYPE) over an ``ow_host_alloc`` block.  A burst
        that a big pool applies on the device is uploaded straight from it (no staging copy).  Free with ``free_host_events``."""
        dt = np.dtype(binding.MIDI_DTYPE)
        ptr = self._lib.ow_host_alloc(dt.itemsize * max(int(n_events), 1), int(device))
        if not ptr:
            raise OwError(binding.take_error(self._lib))
        buf = (C.c_char * (dt.itemsize * int(n_events))).from_address(ptr)
        arr = np.frombuffer(buf, dtype=dt)
        self._host_events = getattr(self, "_host_events", {})
        self._host_events[arr.ctypes.data] = ptr
        return arr

    def free_host_events(self, arr, device=0):
        ptr = getattr(self, "_host_events", {}).pop(arr.ctypes.data, None)
        if ptr:
            self._lib.ow_host_free(C.c_void_p(ptr), int(device))

    def free_host_block(self, block, device=0):
        self._lib.ow_host_free(C.c_void_p(block[0]), int(device))

    def stagger_tremolo(self, n_groups):
        """Test / bench hook (openwurli_hip_test.h): cut the pool into ``n_groups`` tremolo phase groups with decorrelated phases."""
        if self._lib.ow_test_pool_stagger_tremolo(self._h, int(n_groups)) != 0:
            raise OwError(binding.take_error(self._lib))

    def tremolo_groups(self):
        return int(self._lib.ow_test_pool_tremolo_groups(self._h))

    def set_switch(self, name, value):
        """Test / bench hook: change one latched OW_* switch of this pool (openwurli_hip_test.h)."""
        if self._lib.ow_test_pool_set_switch(self._h, name.encode(), int(value)) != 0:
            raise OwError(f"unknown or creation-only switch {name!r}")

    def get_switch(self, name):
        return int(self._lib.ow_test_pool_get_switch(self._h, name.encode()))

    def trajectory_info(self):
        """(engines on the shared tremolo trajectory, samples its store holds, store capacity)."""
        out = (C.c_uint64 * 3)()
        self._lib.ow_test_pool_trajectory_info(self._h, out)
        return int(out[0]), int(out[1]), int(out[2])

    def trajectory_state(self):
        """Test hook: the store behind the pool, in samples -- dict(enqueued, complete, buffers, capacity, oldest_t)."""
        out = (C.c_uint64 * 5)()
        self._lib.ow_test_pool_trajectory_state(self._h, out)
        return dict(zip(("enqueued", "complete", "buffers", "capacity", "oldest_t"), (int(x) for x in out)))

    def midi(self, events):
        """Apply a numpy structured array of events (dtype binding.MIDI_DTYPE) in order."""
        ev = np.ascontiguousarray(events, dtype=np.dtype(binding.MIDI_DTYPE))
        self._lib.ow_pool_midi(self._h, ev.ctypes.data_as(C.c_void_p), ev.size)

    def voice_sum(self, length):
        out = np.zeros((self.n, int(length)), dtype=np.float64)
        if self._lib.ow_pool_read_voice_sum(self._h, out.ctypes.data_as(C.c_void_p), int(length), int(length)) != 0:
            raise OwError(binding.take_error(self._lib))
        return out

    def tremolo_r(self, n_os):
        """CdS-cell resistance stream of the last block, float64 [n, n_os]."""
        out = np.zeros((self.n, int(n_os)), dtype=np.float64)
        if self._lib.ow_pool_read_tremolo_r(self._h, out.ctypes.data_as(C.c_void_p), int(n_os), int(n_os)) != 0:
            raise OwError(binding.take_error(self._lib))
        return out

    def preamp_out(self, n_os):
        out = np.zeros((self.n, int(n_os)), dtype=np.float64)
        if self._lib.ow_pool_read_preamp_out(self._h, out.ctypes.data_as(C.c_void_p), int(n_os), int(n_os)) != 0:
            raise OwError(binding.take_error(self._lib))
        return out

    def power_amp_out(self, n_os):
        """Test tap (openwurli_hip_test.h): the melange power amp's output per chain-rate sample of the last block, float64 [n, n_os]."""
        out = np.zeros((self.n, int(n_os)), dtype=np.float64)
        if self._lib.ow_test_pool_read_power_amp_out(self._h, out.ctypes.data_as(C.c_void_p), int(n_os), int(n_os)) != 0:
            raise OwError(binding.take_error(self._lib) or "power-amp tap not enabled")
        return out

    def power_amp_passes(self):
        """Newton passes every engine's melange power amp spent on the last rendered block (test hook; uint32 per engine)."""
        out = np.zeros(self.n, dtype=np.uint32)
        if self._lib.ow_test_pool_power_amp_passes(self._h, out.ctypes.data_as(C.c_void_p), out.size) != 0:
            raise OwError("ow_test_pool_power_amp_passes failed (no melange power amp in this pool?)")
        return out

    def enable_power_amp_tap(self):
        if self._lib.ow_test_pool_enable_power_amp_tap(self._h) != 0:
            raise OwError("power-amp tap: not a melange power-amp pool")

    def set_profiling(self, on):
        self._lib.ow_pool_set_profiling(self._h, 1 if on else 0)

    def last_kernel_ms(self):
        ms = (C.c_float * 5)()
        self._lib.ow_pool_last_kernel_ms(self._h, ms)
        return dict(zip(("ops", "voices", "tremolo", "preamp", "post"), [float(x) for x in ms]))

    def device_output(self):
        stride = C.c_size_t(0)
        ptr = self._lib.ow_pool_device_output(self._h, C.byref(stride))
        return ptr, stride.value

    def last_block(self):
        """The block the last render left in HBM, copied out: float32 [n, length] (test hook ow_test_device_read)."""
        ptr, stride = self.device_output()
        out = np.zeros((self.n, int(stride)), dtype=np.float32)
        if out.size and self._lib.ow_test_device_read(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, self.device) != 0:
            raise OwError("ow_test_device_read failed")
        return out


def tremolo_prefetch(sample_rate, seconds, device=0):
    """``ow_tremolo_prefetch``: make the first ``seconds`` of the shared Twin-T / CdS trajectory of this host rate exist in HBM now.
    Returns the number of chain-rate samples the store holds."""
    lib = binding.load_library()
    n = lib.ow_tremolo_prefetch(float(sample_rate), int(device), float(seconds))
    if n < 0:
        raise OwError(binding.take_error(lib))
    return int(n)


def tremolo_export(sample_rate, path, device=0):
    """``ow_tremolo_export``: write the trajectory store of this host rate (cut to a checkpoint boundary) to ``path``.  Returns the number
    of chain-rate samples written."""
    lib = binding.load_library()
    n = lib.ow_tremolo_export(float(sample_rate), int(device), str(path).encode())
    if n < 0:
        raise OwError(binding.take_error(lib))
    return int(n)


def tremolo_import(sample_rate, path, device=0):
    """``ow_tremolo_import``: load a file written by :func:`tremolo_export` into the store of this host rate, after the library has checked
    that it would have produced the same samples itself (build id, rate, constants, checksum, two regenerated checkpoint segments).
    Returns the samples taken from the file (0: the store already held more); raises OwError when the file is rejected."""
    lib = binding.load_library()
    n = lib.ow_tremolo_import(float(sample_rate), int(device), str(path).encode())
    if n < 0:
        raise OwError(binding.take_error(lib))
    return int(n)


def tremolo_configure(capacity_seconds=0.0, lead_seconds=-1.0, device=0):
    """``ow_tremolo_configure``: how old an engine may grow on the shared trajectory (seconds since its new / reset; <= 0: the default,
    1 800) and how far the store runs ahead of its oldest reader in the background (seconds; < 0: the default, 60)."""
    lib = binding.load_library()
    if lib.ow_tremolo_configure(int(device), float(capacity_seconds), float(lead_seconds)) != 0:
        raise OwError(binding.take_error(lib))


def render_note(midi_note, velocity, duration_secs, sample_rate, device=0, displacement_scale=None):
    """``Voice::render_note`` (voice.rs:191-199): one voice, reed + pickup only, float64.  ``displacement_scale`` given =
    ``Voice::render_note_with_scale(.., Some(scale))`` (voice.rs:201-221)."""
    lib = binding.load_library()
    n = int(duration_secs * sample_rate)
    out = np.zeros(max(n, 1), dtype=np.float64)
    if displacement_scale is None:
        got = lib.ow_render_note(int(midi_note) & 0xFF, float(velocity), float(duration_secs), float(sample_rate), int(device),
                                 out.ctypes.data_as(C.c_void_p), out.size)
    else:
        got = lib.ow_render_note_with_scale(int(midi_note) & 0xFF, float(velocity), float(duration_secs), float(sample_rate),
                                            float(displacement_scale), int(device), out.ctypes.data_as(C.c_void_p), out.size)
    if got < 0:
        raise OwError(binding.take_error(lib))
    return out[:got]


def normalize_scale(samples):
    """``--normalize`` of ``preamp-bench render`` (main.rs:505-511): the factor its WAV writer applies (0.7 / peak above a 0.7 peak)."""
    lib = binding.load_library()
    a = np.ascontiguousarray(samples, dtype=np.float64)
    return float(lib.ow_normalize_scale(a.ctypes.data_as(C.c_void_p), a.size))


def batch_render(jobs, sample_rate=44100.0, duration_s=2.0, device=0, preamp_kind=0, out_device_ptr=None, stride=None, power_amp_kind=0,
                 no_rail_sag=False):
    """``preamp-bench render`` for a list of jobs (tools/preamp-bench/src/main.rs:371-549), lane = job on the GPU.

    ``jobs``: iterable of dicts with keys note, velocity (0..127) and optional mlp, poweramp, volume, speaker, r_ldr
    (defaults = what ml/render_model_notes.py:57-73 passes: --volume 1.0 --no-poweramp --no-mlp --speaker 0.0, LDR 1 Mohm).
    Returns float64 [n_jobs, n]; with ``out_device_ptr`` the result stays in HBM at that address and None is returned.
    """
    lib = binding.load_library()
    jobs = list(jobs)
    arr = (binding.OwJob * len(jobs))()
    for i, j in enumerate(jobs):
        arr[i].note = int(j["note"]); arr[i].velocity = int(j["velocity"])
        arr[i].mlp = 1 if j.get("mlp", False) else 0
        arr[i].poweramp = 1 if j.get("poweramp", False) else 0
        arr[i].volume = float(j.get("volume", 1.0)); arr[i].speaker = float(j.get("speaker", 0.0)); arr[i].r_ldr = float(j.get("r_ldr", 1e6))
        # the command's remaining flags: --tremolo-depth, --no-preamp, --no-attack-noise, --displacement-scale
        arr[i].tremolo_depth = float(j.get("tremolo_depth", 0.0))
        arr[i].no_preamp = 1 if j.get("no_preamp", False) else 0
        arr[i].no_attack_noise = 1 if j.get("no_attack_noise", False) else 0
        ds = j.get("displacement_scale")
        arr[i].has_displacement_scale = 0 if ds is None else 1
        arr[i].displacement_scale = 0.0 if ds is None else float(ds)
    cfg = binding.OwBatchCfg(float(sample_rate), float(duration_s), int(device), int(preamp_kind), int(power_amp_kind), 1 if no_rail_sag else 0)
    n = int(duration_s * sample_rate)
    stride = int(stride or n)
    if out_device_ptr is not None:
        got = lib.ow_batch_render(arr, len(jobs), C.byref(cfg), C.c_void_p(out_device_ptr), stride, 1)
        if got < 0:
            raise OwError(binding.take_error(lib))
        return None
    out = np.zeros((len(jobs), stride), dtype=np.float64)
    got = lib.ow_batch_render(arr, len(jobs), C.byref(cfg), out.ctypes.data_as(C.c_void_p), stride, 0)
    if got < 0:
        raise OwError(binding.take_error(lib))
    return out[:, :got]
