"""The C-ABI library loads and exports every symbol include/openwurli_hip.h declares; without a GPU the product
fails loudly instead of falling back to a CPU path.  No compute calls here (CPU-only container)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(name="openwurli_hip.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ow_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from openwurli_amd import binding
    assert _header_functions() == sorted(binding.SYMBOLS)
    assert _header_functions("openwurli_hip_test.h") == sorted(binding.TEST_SYMBOLS)
    # the drop-in header carries no test or debug hook
    assert not [f for f in _header_functions() if f.startswith(("ow_test_", "ow_debug_"))]


def test_library_exports_every_declared_symbol(hiplib):
    for name in _header_functions() + _header_functions("openwurli_hip_test.h"):
        assert getattr(hiplib, name) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "openwurli_amd", "lib", "libopenwurli_hip.so")], text=True)
    exported = set(re.findall(r" T (ow_[a-z0-9_]+)", out))
    assert set(_header_functions()) | set(_header_functions("openwurli_hip_test.h")) <= exported


def test_headers_compile_as_c99():
    for h in ("openwurli_hip.h", "openwurli_hip_test.h"):
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", h)])


def test_library_contains_gfx950_code_object():
    path = os.path.join(ROOT, "openwurli_amd", "lib", "libopenwurli_hip.so")
    data = open(path, "rb").read()
    assert b"gfx950" in data
    for k in (b"k_voice", b"k_tremolo", b"k_preamp", b"k_post", b"k_apply_ops"):
        assert k in data


def test_no_cpu_fallback_without_a_device(hiplib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is exercised on CPU-only hosts")
    import openwurli_amd as ow
    with pytest.raises(ow.OwError):
        ow.WurliEngine(48000.0)
    with pytest.raises(ow.OwError):
        ow.render_note(60, 0.8, 0.1, 48000.0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "openwurli_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle/" not in text and "oracle_binding" not in text and "libow_oracle" not in text, os.path.join(dirpath, f)


def test_abi_version_and_struct_size_guards(hiplib):
    """the by-pointer configuration structs were widened in place.  The header now carries OW_ABI_VERSION (the binding checks
    it when it loads the library) and the structs carry their own size: a caller built against another header is refused before any
    field is read -- no device needed to see that."""
    import ctypes as C
    import re
    from openwurli_amd import binding
    hdr = open(os.path.join(ROOT, "include", "openwurli_hip.h")).read()
    assert int(re.search(r"#define OW_ABI_VERSION (\d+)", hdr).group(1)) == binding.ABI_VERSION == hiplib.ow_abi_version()
    jobs = (binding.OwJob * 1)(binding.OwJob(note=60, velocity=100, volume=1.0, r_ldr=1e6))
    out = (C.c_double * 16)()
    for field, bad in (("struct_size", C.sizeof(binding.OwBatchCfg) - 8), ("job_size", C.sizeof(binding.OwJob) - 16)):
        cfg = binding.OwBatchCfg(44100.0, 0.0001)
        setattr(cfg, field, bad)
        hiplib.ow_clear_error()
        assert hiplib.ow_batch_render(jobs, 1, C.byref(cfg), C.cast(out, C.c_void_p), 16, 0) < 0
        assert b"ABI mismatch" in hiplib.ow_last_error()
    mcfg = binding.OwMidiRenderCfg()
    mcfg.struct_size = 8
    offs = (C.c_size_t * 2)(0, 0)
    stats = binding.OwMidiRenderStats()
    hiplib.ow_clear_error()
    assert hiplib.ow_render_midi(None, C.cast(offs, C.c_void_p), 1, C.byref(mcfg), None, 0, C.byref(stats)) < 0
    assert b"ABI mismatch" in hiplib.ow_last_error()
    hiplib.ow_clear_error()
