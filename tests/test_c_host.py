"""The C99 host program tests/c/plugin_process.c: the reference's nih-plug shell (plugin/src/lib.rs:36-166) restated in plain C
over include/openwurli_hip.h -- the compiled stand-in for the Rust facade (no Rust toolchain here).

CPU: it compiles as strict C99 against the two headers, links against the library, and fails loudly without a device.
GPU: (1) its process() loop, fed a played script with sample-accurate events, parameter automation, a saved-session parameter
set restored before initialize() and a mid-stream reset(), matches the oracle driven through the same loop; (2) the realtime
path makes NO heap allocation and NO HIP allocation call over 200 renders that include a whole-keyboard re-strike (SURVEY 8b:
nih-plug assert_process_allocs); (3) a failing render hands back silence in every row (fault injection).
"""
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "plugin_process.c")
EXE = os.path.join(ROOT, "tests", "c", "_build", "plugin_process")
LIBDIR = os.path.join(ROOT, "openwurli_amd", "lib")


def build_c_host():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(LIBDIR, "libopenwurli_hip.so"))):
        return EXE
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-rdynamic", "-o", EXE, SRC,
                           "-L" + LIBDIR, "-lopenwurli_hip", "-ldl", "-lm", "-Wl,-rpath," + LIBDIR])
    return EXE


def _run(*args, check=True):
    r = subprocess.run([build_c_host()] + [str(a) for a in args], capture_output=True, text=True, timeout=900)
    if check:
        assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return r


def test_c_host_compiles_as_c99_and_links(hiplib):
    if os.path.exists(EXE):
        os.remove(EXE)
    build_c_host()
    out = subprocess.check_output(["nm", "-D", "--undefined-only", EXE], text=True)
    used = set(re.findall(r"U (ow_[a-z0-9_]+)", out))
    # what the shell needs of the WurliEngine API (plugin/src/lib.rs:27,36-62,96-105,145-146) + the pool entry points of the audit
    for f in ("ow_engine_new", "ow_engine_set_sample_rate", "ow_engine_ensure_buffer_capacity", "ow_engine_reset", "ow_engine_set_volume",
              "ow_engine_set_tremolo_depth", "ow_engine_set_speaker_character", "ow_engine_set_mlp_enabled", "ow_engine_set_noise_enabled",
              "ow_engine_set_noise_gain", "ow_engine_note_on", "ow_engine_note_off", "ow_engine_set_sustain", "ow_engine_render",
              "ow_pool_new", "ow_pool_render", "ow_pool_midi"):
        assert f in used, f


def test_library_imports_no_malloc_family_symbol(hiplib):
    """Every heap allocation of the library's own code goes through operator new (which the audit counts by caller)."""
    out = subprocess.check_output(["nm", "-D", "--undefined-only", os.path.join(LIBDIR, "libopenwurli_hip.so")], text=True)
    syms = set(re.findall(r"U ([A-Za-z_][A-Za-z0-9_]*)", out))
    assert not syms & {"malloc", "calloc", "realloc", "posix_memalign", "aligned_alloc", "memalign", "valloc", "strdup"}


def test_c_host_fails_loudly_without_a_device(hiplib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _run("audit", 2, 4, check=False)
    assert r.returncode == 3 and "ow_pool_new failed" in r.stderr


# ---------------------------------------------------------------------------------------------------------------- GPU
def _script(rng, n_buffers, buf):
    """A played part: chords, repeated notes, pedal, automation, events at offset 0 / mid-buffer / last sample, one reset."""
    lines = [(0, 0, 5, 0, 0.8), (0, 0, 5, 1, 0.9), (0, 0, 5, 2, 0.35)]        # restored session: volume 0.8, tremolo 0.9, speaker 0.35
    held = []
    for b in range(n_buffers):
        evs = []
        if b == n_buffers // 2:
            lines.append((b, 0, 3, 0, 0.45))                                     # automation lands, then the host resets the plugin
            lines.append((b, 0, 4, 0, 0.0))
            held = []
        if b % 7 == 3:
            lines.append((b, 0, 3, int(rng.integers(0, 3)), float(np.float32(rng.uniform(0.1, 1.0)))))
        for _ in range(int(rng.integers(0, 4))):
            t = int(rng.choice([0, 1, buf // 3, buf // 2, buf - 1, int(rng.integers(0, buf))]))
            r = rng.random()
            if r < 0.55:
                note = int(rng.integers(30, 100))                                # incl. out-of-range keys (clamped to 33..96)
                evs.append((t, 0, note, float(np.float32(rng.uniform(0.2, 1.0)))))
                held.append(note)
            elif r < 0.85 and held:
                evs.append((t, 1, held.pop(int(rng.integers(0, len(held)))), 0.0))
            else:
                evs.append((t, 2, 64, float(rng.integers(0, 2))))
        for (t, ty, note, val) in sorted(evs, key=lambda e: e[0]):             # nih-plug delivers events in timing order
            lines.append((b, t, ty, note, val))
    return [(b, t, ty, note, float(np.float32(val))) for (b, t, ty, note, val) in lines]     # the C side reads every value as f32


def _oracle_plugin(ob, lines, sr, buf, n_buffers):
    """The same shell over the oracle engine (lib.rs:36-166 restated once more, in Python)."""
    params = {"volume": 0.5, "tremolo_depth": 0.5, "speaker_character": 0.0}
    names = ["volume", "tremolo_depth", "speaker_character"]
    for (_, _, ty, note, val) in lines:
        if ty == 5:
            params[names[note]] = val
    e = ob.OracleEngine(44100.0)

    def sync():
        e.set_volume(params["volume"]); e.set_tremolo_depth(params["tremolo_depth"]); e.set_speaker_character(params["speaker_character"])
        e.set_mlp_enabled(True); e.set_noise_enabled(False); e.set_noise_gain(1.0)
    e.set_sample_rate(sr); sync(); e.reset()
    out = []
    li = 0
    lines = [l for l in lines if l[2] != 5]
    for b in range(n_buffers):
        evs = []
        while li < len(lines) and lines[li][0] == b:
            _, t, ty, note, val = lines[li]
            if ty == 3:
                params[names[note]] = val
            elif ty == 4:
                e.reset()
            else:
                evs.append((t, ty, note, val))
            li += 1
        sync()
        start, nxt = 0, 0
        block = np.zeros(buf, dtype=np.float32)

        def handle(ev):
            _, ty, note, val = ev
            if ty == 0:
                e.note_on(note & 0xFF, np.float32(val))
            elif ty == 1:
                e.note_off(note & 0xFF)
            elif ty == 2:
                e.set_sustain(np.float32(val) >= 0.5)
        while start < buf:
            while nxt < len(evs) and evs[nxt][0] <= start:
                handle(evs[nxt]); nxt += 1
            end = min(evs[nxt][0], buf) if nxt < len(evs) else buf
            if end > start:
                block[start:end] = e.render(end - start)
            start = end
        while nxt < len(evs):
            handle(evs[nxt]); nxt += 1
        out.append(block)
    n_active = e.active_voice_count()
    e.close()
    return np.concatenate(out), n_active


@pytest.mark.gpu
@pytest.mark.parametrize("sr,buf", [(48000.0, 256), (44100.0, 333)])
def test_c_host_process_loop_matches_the_oracle(hiplib, oracle, tmp_path, sr, buf):
    rng = np.random.default_rng(int(sr) + buf)
    n_buffers = 60
    lines = _script(rng, n_buffers, buf)
    sp = tmp_path / "script.txt"
    sp.write_text("".join(f"{b} {t} {ty} {note} {val!r}\n" for (b, t, ty, note, val) in lines))
    op = tmp_path / "out.f32"
    r = _run("render", sp, op, sr, buf, n_buffers)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    g = np.fromfile(op, dtype="<f4")
    assert g.size == n_buffers * buf
    c, n_active = _oracle_plugin(oracle, lines, sr, buf, n_buffers)
    rep = oracle.parity_report(g, c, abs_floor=oracle.ABS_FLOOR_DENSE)
    assert rep["peak"] > 1e-2 and rep["n_bad"] == 0, rep
    assert info["active_voices"] == n_active and info["nan_guard_fires"] == 0 and info["last_error"] == ""


@pytest.mark.gpu
@pytest.mark.parametrize("n_engines", [1, 300, 16384])
def test_realtime_path_does_not_allocate(hiplib, n_engines):
    """1 = the pool-of-one a plugin instance is; 300 = single-threaded host paths; 16384 = the sliced host paths on the worker threads."""
    r = _run("audit", n_engines, 200, check=False)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["lib_operator_new"] == 0 and info["hip_alloc_calls"] == 0, info
    assert r.returncode == 0 and info["last_error"] == "" and info["peak"] > 1e-3, (info, r.stderr[-1000:])


@pytest.mark.gpu
@pytest.mark.parametrize("n_engines", [1, 70])
def test_failed_render_degrades_to_silence_in_every_row(hiplib, n_engines):
    r = _run("fault", n_engines, check=False)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0, (info, r.stderr[-1000:])
    assert info["rows_not_silent"] == 0 and info["padding_untouched"] == 1 and info["error_reported"] == 1
