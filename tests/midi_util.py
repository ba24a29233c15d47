"""Hand-rolled Standard MIDI File writer + an independent timing model for the render-midi tests (test infrastructure)."""
import struct


def vlq(v):
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    return bytes(reversed(out))


def track_bytes(items, running_status=False):
    """items: (delta_ticks, kind, a, b): kind in 'on', 'off', 'cc', 'pc', 'tempo', 'text', 'sysex', 'bend'."""
    body = b""
    last = None
    for delta, kind, a, b in items:
        body += vlq(delta)
        if kind == "tempo":
            body += b"\xFF\x51\x03" + struct.pack(">I", a)[1:]
            last = None
        elif kind == "text":
            body += b"\xFF\x01" + vlq(len(a)) + a
            last = None
        elif kind == "sysex":
            body += b"\xF0" + vlq(len(a)) + a
            last = None
        else:
            status = {"on": 0x90, "off": 0x80, "cc": 0xB0, "pc": 0xC0, "bend": 0xE0}[kind]
            data = bytes([a]) if kind == "pc" else bytes([a, b])
            if running_status and last == status:
                body += data
            else:
                body += bytes([status]) + data
            last = status
    body += b"\x00\xFF\x2F\x00"
    return b"MTrk" + struct.pack(">I", len(body)) + body


def smf_bytes(tracks, ticks_per_beat=480, fmt=1, running_status=False):
    return b"MThd" + struct.pack(">IHHH", 6, fmt, len(tracks), ticks_per_beat) + b"".join(track_bytes(t, running_status) for t in tracks)


def expected_events(tracks, ticks_per_beat=480, track_filter=None):
    """(time_s, type, note, value) per the command's rules (main.rs:1651-1708): per-track tempo, vel-0 note-on = note-off, CC64."""
    out = []
    for ti, items in enumerate(tracks):
        tempo, t = 500000.0, 0.0
        emit = track_filter is None or track_filter == ti
        for delta, kind, a, b in items:
            t += (float(delta) / float(ticks_per_beat)) * (tempo / 1000000.0)
            if kind == "tempo":
                tempo = float(a)
            elif not emit:
                continue
            elif kind == "on":
                out.append((t, 1, a, 0) if b == 0 else (t, 0, a, b))
            elif kind == "off":
                out.append((t, 1, a, 0))
            elif kind == "cc" and a == 64:
                out.append((t, 2, 0, 1 if b >= 64 else 0))
    return out
