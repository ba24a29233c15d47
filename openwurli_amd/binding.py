"""ctypes binding of include/openwurli_hip.h (the C-ABI of the HIP library)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OwError(RuntimeError):
    pass


class OwDiag(C.Structure):
    _fields_ = [
        ("active_voices", C.c_uint32), ("held_voices", C.c_uint32), ("sustained_voices", C.c_uint32),
        ("releasing_voices", C.c_uint32), ("steal_voices", C.c_uint32), ("sustain_held", C.c_uint32),
        ("nan_guard_fires", C.c_uint64), ("tremolo_be_fallbacks", C.c_uint64),
        ("preamp_nan_resets", C.c_uint64), ("output_nan_resets", C.c_uint64),
    ]


class OwPowerAmpDiag(C.Structure):
    _fields_ = [("clamp_count", C.c_uint64), ("nr_max_iter_count", C.c_uint64), ("peak_output_volts", C.c_double),
                ("nan_resets", C.c_uint64), ("guard_resets", C.c_uint64), ("rail_pos_volts", C.c_double), ("rail_neg_volts", C.c_double)]


class OwJob(C.Structure):
    _fields_ = [("note", C.c_uint8), ("velocity", C.c_uint8), ("mlp", C.c_uint8), ("poweramp", C.c_uint8),
                ("no_preamp", C.c_uint8), ("no_attack_noise", C.c_uint8), ("has_displacement_scale", C.c_uint8), ("reserved", C.c_uint8),
                ("volume", C.c_double), ("speaker", C.c_double), ("r_ldr", C.c_double),
                ("tremolo_depth", C.c_double), ("displacement_scale", C.c_double)]


class OwMidiEvent(C.Structure):
    _fields_ = [("engine", C.c_uint32), ("type", C.c_uint8), ("note", C.c_uint8), ("reserved", C.c_uint16), ("value", C.c_float)]


MIDI_DTYPE = [("engine", "<u4"), ("type", "u1"), ("note", "u1"), ("reserved", "<u2"), ("value", "<f4")]


ABI_VERSION = 4      # include/openwurli_hip.h OW_ABI_VERSION; load_library() checks it against ow_abi_version()


class OwBatchCfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("job_size", C.c_uint32),
                ("sample_rate", C.c_double), ("duration_s", C.c_double), ("device", C.c_int), ("preamp_kind", C.c_int),
                ("power_amp_kind", C.c_int), ("no_rail_sag", C.c_int)]

    def __init__(self, sample_rate=44100.0, duration_s=2.0, device=0, preamp_kind=0, power_amp_kind=0, no_rail_sag=0):
        super().__init__(C.sizeof(OwBatchCfg), C.sizeof(OwJob), sample_rate, duration_s, device, preamp_kind, power_amp_kind, no_rail_sag)


class OwAliasAuditResult(C.Structure):
    _fields_ = [("f0_hz", C.c_double), ("h1_dbfs", C.c_double), ("harmonic_db", C.c_double * 12), ("harmonic_dbc", C.c_double * 12),
                ("max_step_up_db", C.c_double), ("max_step_up_from_harmonic", C.c_uint32), ("reserved", C.c_uint32),
                ("hf_band_dbc", C.c_double)]


class OwTimedEvent(C.Structure):
    _fields_ = [("time_s", C.c_double), ("type", C.c_uint8), ("note", C.c_uint8), ("value", C.c_uint8), ("reserved", C.c_uint8 * 5)]


TIMED_EVENT_DTYPE = [("time_s", "<f8"), ("type", "u1"), ("note", "u1"), ("value", "u1"), ("reserved", "u1", (5,))]


class OwMidiRenderCfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
                ("volume", C.c_double), ("speaker", C.c_double), ("tail_s", C.c_double), ("no_poweramp", C.c_int), ("device", C.c_int),
                ("preamp_kind", C.c_int), ("power_amp_kind", C.c_int), ("no_rail_sag", C.c_int), ("reserved", C.c_int)]

    def __init__(self, volume=0.6, speaker=1.0, tail_s=2.0, no_poweramp=0, device=0, preamp_kind=0, power_amp_kind=0, no_rail_sag=0, reserved=0):
        super().__init__(C.sizeof(OwMidiRenderCfg), 0, volume, speaker, tail_s, no_poweramp, device, preamp_kind, power_amp_kind, no_rail_sag, reserved)


class OwMidiRenderStats(C.Structure):
    _fields_ = [("n_samples", C.c_uint64), ("note_ons", C.c_uint64), ("peak_polyphony", C.c_uint64)]


# every symbol include/openwurli_hip.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
class OwSegment(C.Structure):
    _fields_ = [("row", C.c_uint32), ("start", C.c_uint32), ("end", C.c_uint32), ("n_harmonics", C.c_uint32), ("f0", C.c_double)]


SEGMENT_DTYPE = [("row", "<u4"), ("start", "<u4"), ("end", "<u4"), ("n_harmonics", "<u4"), ("f0", "<f8")]
MAX_HARMONICS = 8
WAV_NONE, WAV_ROUND, WAV_TRUNCATE = -1, 0, 1

SYMBOLS = {
    "ow_abi_version": (C.c_int, []),
    "ow_last_error": (C.c_char_p, []),
    "ow_clear_error": (None, []),
    "ow_pool_new": (_VP, [C.c_double, C.c_size_t, C.c_int, C.c_int]),
    "ow_pool_new_with": (_VP, [C.c_double, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "ow_pool_new_kinds": (_VP, [C.c_double, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ow_pool_free": (None, [_VP]),
    "ow_pool_size": (C.c_size_t, [_VP]),
    "ow_pool_engine": (_VP, [_VP, C.c_size_t]),
    "ow_engine_pool": (_VP, [_VP]),
    "ow_pool_set_sample_rate": (C.c_int, [_VP, C.c_double]),
    "ow_pool_reset": (None, [_VP]),
    "ow_pool_ensure_buffer_capacity": (None, [_VP, C.c_size_t]),
    "ow_pool_render": (None, [_VP, _VP, C.c_size_t, C.c_size_t]),
    "ow_pool_midi": (None, [_VP, _VP, C.c_size_t]),
    "ow_pool_device_output": (_VP, [_VP, C.POINTER(C.c_size_t)]),
    "ow_pool_read_voice_sum": (C.c_int, [_VP, _VP, C.c_size_t, C.c_size_t]),
    "ow_pool_read_preamp_out": (C.c_int, [_VP, _VP, C.c_size_t, C.c_size_t]),
    "ow_pool_read_tremolo_r": (C.c_int, [_VP, _VP, C.c_size_t, C.c_size_t]),
    "ow_tremolo_prefetch": (C.c_longlong, [C.c_double, C.c_int, C.c_double]),
    "ow_tremolo_export": (C.c_longlong, [C.c_double, C.c_int, C.c_char_p]),
    "ow_tremolo_import": (C.c_longlong, [C.c_double, C.c_int, C.c_char_p]),
    "ow_tremolo_configure": (C.c_int, [C.c_int, C.c_double, C.c_double]),
    "ow_pool_stream": (_VP, [_VP]),
    "ow_pool_set_profiling": (None, [_VP, C.c_int]),
    "ow_pool_last_kernel_ms": (None, [_VP, C.POINTER(C.c_float)]),
    "ow_engine_new": (_VP, [C.c_double, C.c_int, C.c_int]),
    "ow_engine_new_with": (_VP, [C.c_double, C.c_int, C.c_int, C.c_int]),
    "ow_engine_new_kinds": (_VP, [C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ow_engine_set_rail_sag": (None, [_VP, C.c_int]),
    "ow_engine_rail_sag_enabled": (C.c_int, [_VP]),
    "ow_engine_power_amp_diag": (None, [_VP, C.POINTER(OwPowerAmpDiag)]),
    "ow_engine_free": (None, [_VP]),
    "ow_engine_set_sample_rate": (None, [_VP, C.c_double]),
    "ow_engine_reset": (None, [_VP]),
    "ow_engine_warm_up": (None, [_VP]),
    "ow_engine_ensure_buffer_capacity": (None, [_VP, C.c_size_t]),
    "ow_engine_note_on": (None, [_VP, C.c_uint8, C.c_float]),
    "ow_engine_note_off": (None, [_VP, C.c_uint8]),
    "ow_engine_set_sustain": (None, [_VP, C.c_int]),
    "ow_engine_set_volume": (None, [_VP, C.c_double]),
    "ow_engine_set_tremolo_depth": (None, [_VP, C.c_double]),
    "ow_engine_set_speaker_character": (None, [_VP, C.c_double]),
    "ow_engine_set_mlp_enabled": (None, [_VP, C.c_int]),
    "ow_engine_set_noise_enabled": (None, [_VP, C.c_int]),
    "ow_engine_set_noise_gain": (None, [_VP, C.c_double]),
    "ow_engine_set_noise_seed": (None, [_VP, C.c_uint64]),
    "ow_engine_render": (None, [_VP, _VP, C.c_size_t]),
    "ow_engine_get_diag": (None, [_VP, C.POINTER(OwDiag)]),
    "ow_engine_slot_state": (C.c_int, [_VP, C.c_int]),
    "ow_engine_slot_note": (C.c_int, [_VP, C.c_int]),
    "ow_engine_has_steal_voice_for": (C.c_int, [_VP, C.c_uint8]),
    "ow_render_note": (C.c_longlong, [C.c_uint8, C.c_double, C.c_double, C.c_double, C.c_int, _VP, C.c_size_t]),
    "ow_render_note_with_scale": (C.c_longlong, [C.c_uint8, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _VP, C.c_size_t]),
    "ow_normalize_scale": (C.c_double, [_VP, C.c_size_t]),
    "ow_batch_render": (C.c_longlong, [C.POINTER(OwJob), C.c_size_t, C.POINTER(OwBatchCfg), _VP, C.c_size_t, C.c_int]),
    "ow_device_alloc": (_VP, [C.c_size_t, C.c_int]),
    "ow_device_free": (None, [_VP, C.c_int]),
    "ow_host_alloc": (_VP, [C.c_size_t, C.c_int]),
    "ow_host_free": (None, [_VP, C.c_int]),
    "ow_wav24_quantize": (C.c_int, [_VP, C.c_size_t, C.c_double, C.c_int, _VP]),
    "ow_wav24_write": (C.c_int, [C.c_char_p, _VP, C.c_size_t, C.c_uint32, C.c_double, C.c_int]),
    "ow_extract_harmonics": (C.c_int, [_VP, C.c_size_t, C.c_size_t, C.c_double, _VP, C.c_size_t, C.c_double, C.c_int, C.c_int, C.c_int,
                                       _VP, _VP, _VP]),
    "ow_alias_audit_analyze": (C.c_int, [_VP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, _VP, C.c_int, C.c_int, _VP]),
    "ow_alias_audit_run": (C.c_int, [_VP, _VP, C.c_size_t, C.c_int, C.c_int, _VP, _VP, C.c_size_t]),
    "ow_smf_parse": (C.c_longlong, [_VP, C.c_size_t, C.c_int, _VP, C.c_size_t]),
    "ow_render_midi": (C.c_longlong, [_VP, _VP, C.c_size_t, _VP, _VP, C.c_size_t, _VP]),
}


# include/openwurli_hip_test.h: test and debug hooks (not part of the drop-in boundary; bound for tests/ only)
TEST_SYMBOLS = {
    "ow_test_engine_new": (_VP, [C.c_double]),
    "ow_test_engine_free": (None, [_VP]),
    "ow_test_engine_take_ops": (C.c_size_t, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t]),
    "ow_test_engine_after_render": (None, [_VP, C.c_size_t, C.c_uint64]),
    "ow_test_engine_masks": (C.c_uint64, [_VP, C.c_int]),
    "ow_debug_mlp_raw": (C.c_int, [_VP, _VP, C.c_size_t, _VP, C.c_int, C.c_int]),
    "ow_debug_div": (C.c_int, [_VP, _VP, C.c_size_t, _VP, _VP, C.c_int]),
    "ow_debug_div_const": (C.c_int, [C.c_int, _VP, C.c_size_t, _VP, _VP, _VP, C.c_int]),
    "ow_debug_div_forms": (C.c_int, [C.c_int, _VP, _VP, _VP, C.c_size_t, _VP, _VP, C.c_int]),
    "ow_debug_unary": (C.c_int, [C.c_int, _VP, C.c_size_t, _VP, _VP, C.c_int]),
    "ow_test_inject_render_faults": (None, [_VP, C.c_int]),
    "ow_debug_power_amp": (C.c_int, [C.c_double, _VP, C.c_size_t, C.c_size_t, C.c_int, _VP, _VP, _VP, _VP, _VP, C.c_int]),
    "ow_test_pool_enable_power_amp_tap": (C.c_int, [_VP]),
    "ow_test_pool_power_amp_passes": (C.c_int, [_VP, _VP, C.c_size_t]),
    "ow_test_pool_read_power_amp_out": (C.c_int, [_VP, _VP, C.c_size_t, C.c_size_t]),
    "ow_test_engine_poke_power_amp_node": (C.c_int, [_VP, C.c_int, C.c_double]),
    "ow_test_pool_stagger_tremolo": (C.c_int, [_VP, C.c_size_t]),
    "ow_test_pool_tremolo_groups": (C.c_size_t, [_VP]),
    "ow_test_clear_settle_caches": (C.c_int, []),
    "ow_test_pool_set_switch": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "ow_test_pool_get_switch": (C.c_int, [_VP, C.c_char_p]),
    "ow_test_pool_trajectory_info": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "ow_test_pool_trajectory_state": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "ow_test_host_melange_paths": (C.c_int, [C.c_double]),
    "ow_test_device_read": (C.c_int, [_VP, _VP, C.c_size_t, C.c_int]),
    "ow_test_engine_poke_voice": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.c_double]),
    "ow_test_engine_poke_preamp_node": (C.c_int, [_VP, C.c_int, C.c_int, C.c_double]),
    "ow_test_engine_read_preamp_state": (C.c_int, [_VP, C.c_int, _VP]),
    "ow_test_host_matrices": (C.c_int, [C.c_int, C.c_double, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "ow_debug_trem_trajectory": (C.c_int, [C.c_double, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int, _VP, _VP, _VP, _VP, _VP, C.c_int, _VP, _VP]),
}


def library_path():
    return os.environ.get("OPENWURLI_HIP_LIB", os.path.join(_HERE, "lib", "libopenwurli_hip.so"))


def load_library():
    """Load the HIP library.  Fails loudly when it has not been built -- there is no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise OwError(f"{path} not found: build it with ./build.sh (hipcc --offload-arch=gfx950); "
                      "openwurli-hip has no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in list(SYMBOLS.items()) + list(TEST_SYMBOLS.items()):
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.ow_abi_version() != ABI_VERSION:
        raise OwError(f"{path} was built against OW_ABI_VERSION {lib.ow_abi_version()}, this binding against {ABI_VERSION}: rebuild with ./build.sh")
    _LIB = lib
    return lib


def last_error(lib=None):
    lib = lib or load_library()
    msg = lib.ow_last_error()
    return msg.decode() if msg else ""


def take_error(lib=None):
    """The recorded error message, cleared (so that a later raise_if_error does not report it again)."""
    lib = lib or load_library()
    msg = last_error(lib)
    lib.ow_clear_error()
    return msg


def raise_if_error(lib=None):
    """The realtime C entry points return void and degrade to silence; the Python mirror turns the recorded reason into an exception
    (the block it returns alongside is the silence the C contract promises)."""
    lib = lib or load_library()
    msg = last_error(lib)
    if msg:
        lib.ow_clear_error()
        raise OwError(msg)
