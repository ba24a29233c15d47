"""`preamp-bench render-midi` (tools/preamp-bench/src/main.rs:1603-1923) over the C-ABI: Standard MIDI Files (or event lists)
rendered through the command's own 64-slot voice manager and output chain, many files per call (one wavefront per file for
the voices, one lane pair per file for the chain)."""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from .binding import OwError, OwMidiRenderCfg, OwMidiRenderStats, TIMED_EVENT_DTYPE, load_library, take_error

BASE_SR = 44100.0        # main.rs:27
NOTE_ON, NOTE_OFF, PEDAL = 0, 1, 2


def _err(L):
    return OwError(take_error(L))


def parse_smf(data: bytes, track: Optional[int] = None) -> np.ndarray:
    """Timed events of a Standard MIDI File in file order (main.rs:1627-1708); `track` = `--track N`."""
    L = load_library()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    tf = -1 if track is None else int(track)
    n = L.ow_smf_parse(buf.ctypes.data_as(C.c_void_p), buf.size, tf, None, 0)
    if n < 0:
        raise _err(L)
    ev = np.zeros(n, dtype=np.dtype(TIMED_EVENT_DTYPE))
    if n and L.ow_smf_parse(buf.ctypes.data_as(C.c_void_p), buf.size, tf, ev.ctypes.data_as(C.c_void_p), n) != n:
        raise _err(L)
    return ev


def events(items: Sequence) -> np.ndarray:
    """[(time_s, type, note, value), ...] -> event array."""
    ev = np.zeros(len(items), dtype=np.dtype(TIMED_EVENT_DTYPE))
    for i, (t, ty, note, value) in enumerate(items):
        ev[i]["time_s"], ev[i]["type"], ev[i]["note"], ev[i]["value"] = t, ty, note, value
    return ev


def render_midi(jobs: Sequence[np.ndarray], volume=0.60, speaker=1.0, no_poweramp=False, tail=2.0, device=0, return_stats=False, preamp_kind=0,
                power_amp_kind=0, no_rail_sag=False):
    """Render every event list of `jobs`; returns a list of f64 arrays (one per job, its own length)."""
    L = load_library()
    jobs = [np.ascontiguousarray(j, dtype=np.dtype(TIMED_EVENT_DTYPE)) for j in jobs]
    offs = np.zeros(len(jobs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([j.size for j in jobs])
    allev = np.concatenate(jobs) if jobs else np.zeros(0, dtype=np.dtype(TIMED_EVENT_DTYPE))
    cfg = OwMidiRenderCfg(float(volume), float(speaker), float(tail), 1 if no_poweramp else 0, int(device), int(preamp_kind), int(power_amp_kind),
                          1 if no_rail_sag else 0, 0)
    stats = (OwMidiRenderStats * max(len(jobs), 1))()
    evp = allev.ctypes.data_as(C.c_void_p) if allev.size else None
    longest = L.ow_render_midi(evp, offs.ctypes.data_as(C.c_void_p), len(jobs), C.byref(cfg), None, 0, C.cast(stats, C.c_void_p))
    if longest < 0:
        raise _err(L)
    out = np.zeros((len(jobs), max(int(longest), 1)))
    if longest > 0 and L.ow_render_midi(evp, offs.ctypes.data_as(C.c_void_p), len(jobs), C.byref(cfg), out.ctypes.data_as(C.c_void_p), out.shape[1],
                                        C.cast(stats, C.c_void_p)) < 0:
        raise _err(L)
    res = [out[j, :int(stats[j].n_samples)].copy() for j in range(len(jobs))]
    if return_stats:
        return res, [(int(stats[j].note_ons), int(stats[j].peak_polyphony)) for j in range(len(jobs))]
    return res


def render_midi_files(paths: Sequence[str], outputs: Optional[Sequence[str]] = None, track: Optional[int] = None, **kw) -> List[np.ndarray]:
    """The command itself for a list of files: parse, render, and (if `outputs`) write 24-bit WAVs like write_wav_24bit(.., 1.0)."""
    from .features import write_wav_24bit
    jobs = [parse_smf(open(p, "rb").read(), track) for p in paths]
    res = render_midi(jobs, **kw)
    if outputs is not None:
        for o, x in zip(outputs, res):
            if x.size:                     # "No note events found": the command writes nothing
                write_wav_24bit(o, x, int(BASE_SR))
    return res
