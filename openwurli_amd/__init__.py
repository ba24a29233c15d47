"""openwurli-hip: MI355X (gfx950) render core for the OpenWurli DSP hot path.

The product is the C-ABI library ``openwurli_amd/lib/libopenwurli_hip.so`` (declared in
``include/openwurli_hip.h``).  This package is a thin ctypes host mirror of the reference's
``WurliEngine`` / ``Voice::render_note`` API over that library.  There is no CPU fallback:
importing works anywhere, constructing an engine without the built library or without a
HIP device raises.
"""
from .binding import load_library, library_path, OwError  # noqa: F401
from .engine import WurliEngine, EnginePool, VoiceState, render_note, batch_render, normalize_scale, tremolo_prefetch, tremolo_configure, tremolo_export, tremolo_import  # noqa: F401
from . import features  # noqa: F401
from . import alias_audit  # noqa: F401
from . import midi_render  # noqa: F401

__all__ = ["load_library", "library_path", "OwError", "WurliEngine", "EnginePool", "VoiceState", "render_note", "batch_render", "normalize_scale", "tremolo_prefetch", "tremolo_configure", "tremolo_export", "tremolo_import", "features", "alias_audit", "midi_render"]
