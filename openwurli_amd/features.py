"""Host mirror of the ML-pipeline stage that follows the batch render (`ml/render_model_notes.py:118-237`,
`ml/goertzel_utils.py:60-129`) over the C-ABI: 24-bit WAV output and harmonic features of rendered notes.

The spectra are evaluated by `ow_extract_harmonics` on the GPU; what stays here is the bookkeeping of the reference
script (window tables, rounding of the reported numbers, the 6-point decay fit, centroids).  No CPU fallback.
"""
import ctypes as C

import numpy as np

from . import binding
from .binding import OwError

# render_model_notes.py imports these from extract_harmonics.py:27-36
WINDOWS = (("attack", 0.000, 0.050), ("early_sustain", 0.050, 0.200), ("sustain", 0.200, 0.800))
DECAY_TIMES = (0.1, 0.3, 0.5, 0.8, 1.0, 1.5)
N_HARMONICS = 8


def midi_to_freq(midi_note):
    """goertzel_utils.py:127-129."""
    return 440.0 * 2.0 ** ((midi_note - 69) / 12.0)


def amps_to_dB(amps, ref=None):
    """goertzel_utils.py:120-124."""
    amps = np.asarray(amps, dtype=np.float64)
    if ref is None:
        ref = max(amps[0], 1e-20)
    return 20.0 * np.log10(np.maximum(amps, 1e-20) / ref)


def quantize_24bit(samples, scale=1.0, mode="round"):
    """int32 sample values the reference's WAV writers would store ("round": preamp-bench, "truncate": reed-renderer)."""
    lib = binding.load_library()
    x = np.ascontiguousarray(samples, dtype=np.float64)
    out = np.zeros(x.size, dtype=np.int32)
    m = binding.WAV_ROUND if mode == "round" else binding.WAV_TRUNCATE
    if lib.ow_wav24_quantize(x.ctypes.data_as(C.c_void_p), x.size, float(scale), m, out.ctypes.data_as(C.c_void_p)) != 0:
        raise OwError(binding.take_error(lib))
    return out.reshape(x.shape)


def write_wav_24bit(path, samples, sample_rate, scale=1.0, mode="round"):
    """`write_wav_24bit(path, samples, sample_rate, scale)` of preamp-bench (main.rs:941) / `write_wav` of reed-renderer."""
    lib = binding.load_library()
    x = np.ascontiguousarray(samples, dtype=np.float64)
    m = binding.WAV_ROUND if mode == "round" else binding.WAV_TRUNCATE
    if lib.ow_wav24_write(str(path).encode(), x.ctypes.data_as(C.c_void_p), x.size, int(sample_rate), float(scale), m) != 0:
        raise OwError(binding.take_error(lib))


_WAV_MODES = {None: binding.WAV_NONE, "round": binding.WAV_ROUND, "truncate": binding.WAV_TRUNCATE}


def extract_segments(audio, sample_rate, segments, search_pct=0.01, device=0, wav24=None, device_audio=None):
    """`extract_harmonics_fft` on many segments in one call.

    audio: float64 [rows, stride]; segments: iterable of (row, start, end, n_harmonics, f0).
    wav24: None analyses the samples as given; "round" / "truncate" analyses what a 24-bit WAV written with that quantiser
    reads back as (the reference script always goes through the file).
    device_audio = (pointer, rows, stride): analyse f64 audio that is already in HBM (e.g. what batch_render left there) instead of
    `audio` (which may then be None).
    Returns (amps [n, 8], freqs [n, 8], rms [n])."""
    lib = binding.load_library()
    if device_audio is None:
        a = np.ascontiguousarray(audio, dtype=np.float64)
        if a.ndim == 1:
            a = a[None, :]
        ptr, rows, stride, on_dev = a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], 0
    else:
        ptr, rows, stride, on_dev = C.c_void_p(int(device_audio[0])), int(device_audio[1]), int(device_audio[2]), 1
    seg = np.array([tuple(s) for s in segments], dtype=np.dtype(binding.SEGMENT_DTYPE))
    n = seg.size
    amps = np.zeros((n, binding.MAX_HARMONICS)); freqs = np.zeros((n, binding.MAX_HARMONICS)); rms = np.zeros(n)
    if n == 0:
        return amps, freqs, rms
    rc = lib.ow_extract_harmonics(ptr, rows, stride, float(sample_rate),
                                  seg.ctypes.data_as(C.c_void_p), n, float(search_pct), _WAV_MODES[wav24], int(device), on_dev,
                                  amps.ctypes.data_as(C.c_void_p), freqs.ctypes.data_as(C.c_void_p), rms.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise OwError(binding.take_error(lib))
    return amps, freqs, rms


def extract_harmonics_fft(signal, sr, f0, n_harmonics=8, search_pct=0.01, device=0):
    """Drop-in for goertzel_utils.extract_harmonics_fft: (amps, freqs) of H1..Hn of one signal."""
    x = np.ascontiguousarray(signal, dtype=np.float64)
    amps, freqs, _ = extract_segments(x[None, :], sr, [(0, 0, x.size, n_harmonics, f0)], search_pct, device)
    return amps[0, :n_harmonics].copy(), freqs[0, :n_harmonics].copy()


def _note_segments(row, n_samples, sr, f0):
    """Segments of one rendered note in the order extract_model_features walks them; None where the script skips one."""
    dur = n_samples / sr
    out = []
    for _, w0, w1 in WINDOWS:                                  # render_model_notes.py:146-157
        end = min(w1, dur)
        a, b = int(w0 * sr), min(int(end * sr), n_samples)
        out.append((row, a, b, N_HARMONICS, f0) if (w0 < end and b - a >= 128) else None)
    for t in DECAY_TIMES:                                      # :170-181
        a, b = int(t * sr), min(int((t + 0.100) * sr), n_samples)
        out.append((row, a, b, 1, f0) if (t < dur - 0.05 and b - a >= 64) else None)
    pe, ss, se = min(int(0.010 * sr), n_samples), int(0.100 * sr), min(int(0.200 * sr), n_samples)   # :199-206
    ok = pe > 0 and se > ss
    out.append((row, 0, pe, 0, f0) if ok else None)
    out.append((row, ss, se, 0, f0) if ok else None)
    return out


def extract_model_features(audio, sr, pairs, device=0, wav24="round", device_audio=None):
    """`extract_model_features(wav_paths, pairs)` (render_model_notes.py:118-237) on renders that are still in memory:
    audio[j] is the note of pairs[j] = (midi, velocity), starting at t = 0.  One GPU call for all notes.
    wav24="round" (default) reproduces the script's WAV round trip (preamp-bench writes, soundfile reads) without the files.
    device_audio = (pointer, rows, stride, n_samples): the renders are still in HBM (see render_and_extract)."""
    if device_audio is None:
        a = np.ascontiguousarray(audio, dtype=np.float64)
        n_samples = a.shape[1]
        dev = None
    else:
        a, n_samples, dev = None, int(device_audio[3]), device_audio[:3]
    plan, flat = [], []
    for j, (midi, _) in enumerate(pairs):
        segs = _note_segments(j, n_samples, sr, midi_to_freq(midi))
        plan.append([None if s is None else len(flat) + sum(1 for t in segs[:i] if t is not None) for i, s in enumerate(segs)])
        flat.extend(s for s in segs if s is not None)
    amps, freqs, rms = extract_segments(a, sr, flat, 0.01, device, wav24, dev)
    features = {}
    for j, (midi, vel) in enumerate(pairs):
        idx = plan[j]
        feat = {"midi_note": midi, "velocity_midi": vel, "f0": midi_to_freq(midi), "duration_s": round(n_samples / sr, 4), "windows": {}}
        for w, (name, _, _) in enumerate(WINDOWS):
            if idx[w] is None:
                feat["windows"][name] = None
                continue
            am, fr = amps[idx[w]], freqs[idx[w]]
            feat["windows"][name] = {"amps_linear": [round(float(x), 8) for x in am],
                                     "amps_dB_rel_H1": [round(float(d), 2) for d in amps_to_dB(am)],
                                     "freqs_hz": [round(float(f), 2) for f in fr]}
        decay = [None if idx[3 + i] is None else round(float(amps[idx[3 + i], 0]), 8) for i in range(len(DECAY_TIMES))]
        pts = [(t, x) for t, x in zip(DECAY_TIMES, decay) if x is not None and x > 1e-15]
        rate = None
        if len(pts) >= 3:
            ts = np.array([p[0] for p in pts]); la = np.log10(np.array([p[1] for p in pts]))
            if np.std(ts) > 0:
                rate = round(float(-20.0 * np.polyfit(ts, la, 1)[0]), 2)
        feat["decay"] = {"times_s": list(DECAY_TIMES), "h1_amps": decay, "decay_rate_dB_s": rate}
        feat["overshoot_dB"] = (round(float(20.0 * np.log10(rms[idx[9]] / rms[idx[10]])), 2)
                                if idx[9] is not None and idx[10] is not None else None)
        for name in ("attack", "sustain"):
            w = feat["windows"].get(name)
            if w is None:
                feat[f"centroid_{name}"] = None
                continue
            am = np.array(w["amps_linear"]); fr = np.array(w["freqs_hz"]); ok = am > 1e-15
            feat[f"centroid_{name}"] = round(float(np.sum(fr[ok] * am[ok]) / np.sum(am[ok])), 1) if np.any(ok) else None
        features[(midi, vel)] = feat
    return features


def render_and_extract(pairs, sample_rate=44100.0, duration_s=2.0, device=0, preamp_kind=0, wav24="round"):
    """Stages 3 + 4 of ml/render_model_notes.py in one go, nothing leaving the GPU in between: `preamp-bench render` of every
    (midi, velocity) pair (ow_batch_render into a device buffer) and `extract_model_features` on that buffer.
    Returns the features dictionary."""
    from .engine import batch_render
    lib = binding.load_library()
    pairs = list(pairs)
    n = int(duration_s * sample_rate)
    buf = lib.ow_device_alloc(8 * len(pairs) * n, int(device))
    if not buf:
        raise OwError(binding.take_error(lib))
    try:
        batch_render([{"note": m, "velocity": v} for m, v in pairs], sample_rate=sample_rate, duration_s=duration_s, device=device,
                     preamp_kind=preamp_kind, out_device_ptr=buf, stride=n)
        return extract_model_features(None, sample_rate, pairs, device, wav24, device_audio=(buf, len(pairs), n, n))
    finally:
        lib.ow_device_free(buf, int(device))
