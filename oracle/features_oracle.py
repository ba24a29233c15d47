"""CPU oracle (numpy) of the ML-pipeline stage that consumes batch renders: 24-bit WAV quantisers and the
harmonic feature extraction of `ml/render_model_notes.py`.  TEST INFRASTRUCTURE ONLY: imported by tests/, never
by the product path (openwurli_amd.features calls the HIP library).

Restates (reference file:line):
  * write_wav_24bit            tools/preamp-bench/src/main.rs:941-957   round-half-away, saturating cast, clamp +-(2^23-1)
  * write_wav (reed-renderer)  tools/reed-renderer/src/main.rs:110-126   clamp to +-1, x (2^23-1), truncate toward zero
  * extract_harmonics_fft      ml/goertzel_utils.py:60-107               Hann window, 4x zero-padded rFFT, peak bin in +-1 %
  * amps_to_dB / midi_to_freq  ml/goertzel_utils.py:120-129
  * extract_model_features     ml/render_model_notes.py:118-237          3 windows x 8 harmonics, 6 decay points, overshoot, centroid
Pinned: tests/golden/harmonics_golden.npz holds outputs of the reference's own extract_harmonics_fft / amps_to_dB /
midi_to_freq (imported from /root/reference/ml in the build container by tests/golden/make_harmonics_golden.py); the
reference has no test of its own for these functions.  WAV container bytes follow hound 3.5.1 (Cargo.lock), which is not
vendored: header layout unpinned, sample payload pinned by the quantiser definitions above.
"""
import numpy as np

WINDOWS = (("attack", 0.000, 0.050), ("early_sustain", 0.050, 0.200), ("sustain", 0.200, 0.800))   # render_model_notes.py:32-36 via extract_harmonics.py:27-31
DECAY_TIMES = (0.1, 0.3, 0.5, 0.8, 1.0, 1.5)                                                       # extract_harmonics.py:34
N_HARMONICS = 8
I24_MAX = (1 << 23) - 1


def quantize_round(samples, scale=1.0):
    """preamp-bench: (sample * scale * max).round() as i32, clamped (Rust round = half away from zero; `as` saturates, NaN -> 0)."""
    x = np.asarray(samples, dtype=np.float64) * scale * float(I24_MAX)
    r = np.where(np.isnan(x), 0.0, np.sign(x) * np.floor(np.abs(x) + 0.5))
    return np.clip(r, -I24_MAX, I24_MAX).astype(np.int32)


def quantize_truncate(samples):
    """reed-renderer: (s.clamp(-1, 1) * (2^23 - 1)) as i32 (truncation toward zero; NaN -> 0)."""
    x = np.asarray(samples, dtype=np.float64)
    c = np.where(np.isnan(x), 0.0, np.clip(x, -1.0, 1.0)) * float(I24_MAX)
    return np.trunc(c).astype(np.int32)


def midi_to_freq(midi):
    return 440.0 * 2.0 ** ((midi - 69) / 12.0)


def amps_to_db(amps):
    amps = np.asarray(amps, dtype=np.float64)
    ref = max(amps[0], 1e-20)
    return 20.0 * np.log10(np.maximum(amps, 1e-20) / ref)


def harmonic_bins(n, sr, f0, n_harmonics=N_HARMONICS, search_pct=0.01):
    """Candidate rFFT bins of each harmonic: (k_lo, k_hi) inclusive, or None when the harmonic is skipped
    (above Nyquist - 100 Hz, or no bin inside the +-search_pct mask).  Uses numpy's own frequency axis arithmetic."""
    nfft = 4 * n
    val = 1.0 / (nfft * (1.0 / sr))            # np.fft.rfftfreq: arange(n//2+1) * val
    nb = nfft // 2 + 1
    out = []
    for h in range(n_harmonics):
        fh = f0 * (h + 1)
        if fh >= sr / 2 - 100:
            out.append(None)
            continue
        f_lo, f_hi = fh * (1.0 - search_pct), fh * (1.0 + search_pct)
        k = int(f_lo / val)
        while k > 0 and (k - 1) * val >= f_lo:
            k -= 1
        while k < nb and k * val < f_lo:
            k += 1
        k_lo = k
        while k < nb and k * val <= f_hi:
            k += 1
        out.append((k_lo, k - 1) if k - 1 >= k_lo else None)
    return out, val


def extract_harmonics(signal, sr, f0, n_harmonics=N_HARMONICS, search_pct=0.01):
    """Peak amplitude and frequency of H1..Hn (goertzel_utils.py:60-107)."""
    x = np.asarray(signal, dtype=np.float64)
    n = x.size
    spec = np.abs(np.fft.rfft(x * np.hanning(n), n=4 * n)) * 2.0 / n / 0.5
    bins, val = harmonic_bins(n, sr, f0, n_harmonics, search_pct)
    amps = np.zeros(n_harmonics)
    freqs = np.zeros(n_harmonics)
    for h, b in enumerate(bins):
        if b is None:
            amps[h], freqs[h] = 1e-20, f0 * (h + 1)
            continue
        k = b[0] + int(np.argmax(spec[b[0]:b[1] + 1]))
        amps[h], freqs[h] = spec[k], k * val
    return amps, freqs


def rms(x):
    x = np.asarray(x, dtype=np.float64)
    return 1e-20 if x.size == 0 else max(float(np.sqrt(np.mean(x ** 2))), 1e-20)


def segments_of(n_samples, sr):
    """Segment list of one rendered note (render_model_notes.py:146-201): (kind, name, start, end, n_harmonics) or None entries."""
    dur = n_samples / sr
    segs = []
    for name, w0, w1 in WINDOWS:
        end = min(w1, dur)
        if w0 >= end:
            segs.append(None)
            continue
        a, b = int(w0 * sr), int(end * sr)
        segs.append(("window", name, a, b, N_HARMONICS) if len(range(a, min(b, n_samples))) >= 128 else None)
    for t in DECAY_TIMES:
        if t >= dur - 0.05:
            segs.append(None)
            continue
        a, b = int(t * sr), min(int((t + 0.100) * sr), n_samples)
        segs.append(("decay", t, a, b, 1) if b - a >= 64 else None)
    return segs


def model_features(audio, sr, midi, vel):
    """One entry of extract_model_features (render_model_notes.py:131-234)."""
    audio = np.asarray(audio, dtype=np.float64)
    f0 = midi_to_freq(midi)
    dur = audio.size / sr
    feat = {"midi_note": midi, "velocity_midi": vel, "f0": f0, "duration_s": round(dur, 4), "windows": {}}
    segs = segments_of(audio.size, sr)
    for (name, _, _), s in zip(WINDOWS, segs[:3]):
        if s is None:
            feat["windows"][name] = None
            continue
        amps, freqs = extract_harmonics(audio[s[2]:s[3]], sr, f0, N_HARMONICS)
        db = amps_to_db(amps)
        feat["windows"][name] = {"amps_linear": [round(float(a), 8) for a in amps],
                                 "amps_dB_rel_H1": [round(float(d), 2) for d in db],
                                 "freqs_hz": [round(float(f), 2) for f in freqs]}
    decay = []
    for s in segs[3:]:
        decay.append(None if s is None else round(float(extract_harmonics(audio[s[2]:s[3]], sr, f0, 1)[0][0]), 8))
    pts = [(t, a) for t, a in zip(DECAY_TIMES, decay) if a is not None and a > 1e-15]
    rate = None
    if len(pts) >= 3:
        ts = np.array([p[0] for p in pts]); la = np.log10(np.array([p[1] for p in pts]))
        if np.std(ts) > 0:
            rate = round(float(-20.0 * np.polyfit(ts, la, 1)[0]), 2)
    feat["decay"] = {"times_s": list(DECAY_TIMES), "h1_amps": decay, "decay_rate_dB_s": rate}
    pe, ss, se = min(int(0.010 * sr), audio.size), int(0.100 * sr), min(int(0.200 * sr), audio.size)
    feat["overshoot_dB"] = round(float(20.0 * np.log10(rms(audio[:pe]) / rms(audio[ss:se]))), 2) if pe > 0 and se > ss else None
    for name in ("attack", "sustain"):
        w = feat["windows"].get(name)
        if w is None:
            feat[f"centroid_{name}"] = None
            continue
        a = np.array(w["amps_linear"]); f = np.array(w["freqs_hz"]); ok = a > 1e-15
        feat[f"centroid_{name}"] = round(float(np.sum(f[ok] * a[ok]) / np.sum(a[ok])), 1) if np.any(ok) else None
    return feat
