"""Click-band alias audit on the device: host mirror of ``crates/openwurli-dsp/src/alias_audit.rs`` over the C-ABI
(``ow_alias_audit_run`` / ``ow_alias_audit_analyze``).  Same constants, function names and result fields as the Rust module;
the stimulus renders stay in HBM between the engine pool and the analysis kernels.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from .binding import OwAliasAuditResult, OwError, load_library, take_error

STIMULUS_NOTE = 84                      # alias_audit.rs:28
STIMULUS_VELOCITY = 120                 # :30
STIMULUS_VOLUME = 0.5                   # :32
STIMULUS_NOTES = (72, 84, 91)           # :45
STIMULUS_SAMPLE_RATE = 44100.0          # :47
STIMULUS_RENDER_SECONDS = 1.5           # :50
STIMULUS_ANALYZE_SECONDS = 0.5          # :53
NUM_HARMONICS = 12                      # :56
PLATEAU_FIRST_HARMONIC = 6              # :58
PLATEAU_LAST_HARMONIC = 11              # :60
HF_BAND_LO_HZ = 5000.0                  # :62
HF_BAND_HI_HZ = 18000.0                 # :64


@dataclass
class AliasAuditResult:                 # alias_audit.rs:68-93
    f0_hz: float
    h1_dbfs: float
    harmonic_db: List[float]
    harmonic_dbc: List[float]
    max_step_up_db: float
    max_step_up_from_harmonic: int
    hf_band_dbc: float


@dataclass
class SweepEntry:                       # alias_audit.rs:111-116
    note: int
    velocity: int
    result: AliasAuditResult


def _unpack(r):
    return AliasAuditResult(r.f0_hz, r.h1_dbfs, list(r.harmonic_db), list(r.harmonic_dbc), r.max_step_up_db,
                            int(r.max_step_up_from_harmonic), r.hf_band_dbc)


def _check(L, rc):
    if rc != 0:
        raise OwError(take_error(L))


def midi_note_hz(note):                 # alias_audit.rs:284-287
    return 440.0 * 2.0 ** ((float(note) - 69.0) / 12.0)


def run_notes(notes: Sequence[int], velocities: Sequence[int], device=0, preamp_kind=0, return_signals=False):
    """``run_with_note`` for many (note, velocity) pairs in one pool (one engine per pair)."""
    L = load_library()
    nn = np.ascontiguousarray(notes, dtype=np.uint8)
    vv = np.ascontiguousarray(velocities, dtype=np.uint8)
    if nn.shape != vv.shape or nn.ndim != 1:
        raise ValueError("notes and velocities must be 1-D and of equal length")
    res = (OwAliasAuditResult * max(nn.size, 1))()
    total = int(STIMULUS_SAMPLE_RATE * STIMULUS_RENDER_SECONDS)
    sig = np.zeros((nn.size, total)) if return_signals else None
    _check(L, L.ow_alias_audit_run(nn.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p), nn.size, int(device), int(preamp_kind),
                                   C.cast(res, C.c_void_p), sig.ctypes.data_as(C.c_void_p) if sig is not None else None, total))
    out = [_unpack(res[k]) for k in range(nn.size)]
    return (out, sig) if return_signals else out


def run_with_note(note, velocity, device=0) -> AliasAuditResult:   # alias_audit.rs:104-108
    return run_notes([note], [velocity], device=device)[0]


def run(device=0) -> AliasAuditResult:                             # alias_audit.rs:97-99
    return run_with_note(STIMULUS_NOTE, STIMULUS_VELOCITY, device=device)


def run_sweep(device=0) -> List[SweepEntry]:                       # alias_audit.rs:123-133
    res = run_notes(STIMULUS_NOTES, [STIMULUS_VELOCITY] * len(STIMULUS_NOTES), device=device)
    return [SweepEntry(n, STIMULUS_VELOCITY, r) for n, r in zip(STIMULUS_NOTES, res)]


def analyze(signals, sr, nominal_f0, device=0, length: Optional[int] = None) -> List[AliasAuditResult]:
    """``analyze`` (alias_audit.rs:163-211) of one signal or a [n][len] stack of signals."""
    L = load_library()
    a = np.ascontiguousarray(np.atleast_2d(np.asarray(signals, dtype=np.float64)))
    f = np.ascontiguousarray(np.atleast_1d(np.asarray(nominal_f0, dtype=np.float64)))
    if f.size != a.shape[0]:
        raise ValueError("one nominal f0 per signal")
    res = (OwAliasAuditResult * max(a.shape[0], 1))()
    _check(L, L.ow_alias_audit_analyze(a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], a.shape[1] if length is None else int(length),
                                       float(sr), f.ctypes.data_as(C.c_void_p), int(device), 0, C.cast(res, C.c_void_p)))
    return [_unpack(res[k]) for k in range(a.shape[0])]
