// openwurli-hip: the voice-pool state machine of one engine (engine.rs:39-62 VoiceSlot, :299-374 note_on / note_off / set_sustain,
// :569-590 allocate_voice, :592-602 cleanup_voices) as plain data + functions that compile for the host AND for the device.
//
// The machine is integer, branchy and needs no audio: it lives on the host (one engine = one OwVm in a pinned array of the pool).  A
// big pool that receives a BURST of events (ow_pool_midi: a whole-keyboard re-strike of 131 072 engines is 16.7 M events) runs the same
// functions on the device instead -- one lane per engine over its slice of the event list (k_vm_events) -- and copies the states back:
// 8 bytes per event up, 912 bytes per engine down, no per-event host work (a 16-thread host spent 25-60 ms on such a burst, a 2-thread
// rank of an 8-GPU node 125 ms, with the GPU idle).  One implementation, so the host-logic tests (tests/test_host_logic.py) cover
// what the device runs, and tests/test_gpu_boundary.py compares the two paths slot for slot.
#pragma once
#include <stdint.h>
#include "ow_types.h"
#include "../../include/openwurli_hip.h"

#if defined(__HIPCC__)
#define OW_VM_HD __host__ __device__ inline
#else
#define OW_VM_HD inline
#endif

#define OW_VM_OPS_MAX 192        // slot ops one engine may queue on the device between two renders: damper + move-to-steal + note-on per key

struct OwVm {
    // slot state as four disjoint bitmasks indexed by OW_VOICE_* (exactly one bit set per slot): note_on / note_off / allocate_voice are
    // ctz / popcount instead of 64-slot walks
    uint64_t st_mask[4];
    uint64_t has_voice, has_steal;      // bit s: slot s holds a voice / a steal voice (Option<Voice> of VoiceSlot)
    uint64_t main_mask, steal_mask;     // bit s: slot s renders a voice / a steal voice (engine.rs:471-493); kept incrementally
    uint64_t age[OW_MAX_VOICES];
    uint32_t steal_fade[OW_MAX_VOICES];
    uint8_t midi_of[OW_MAX_VOICES];
    uint64_t age_counter;
    uint32_t n_dev_ops;                 // ops this engine queued ON THE DEVICE (k_vm_events) that the next render has to apply
    uint8_t sustain_held, mlp_enabled, dev_overflow, pad;
};

OW_VM_HD void vm_init(OwVm& v) {
    v.st_mask[0] = ~0ull; v.st_mask[1] = v.st_mask[2] = v.st_mask[3] = 0ull;
    v.has_voice = v.has_steal = v.main_mask = v.steal_mask = 0ull;
    for (int i = 0; i < OW_MAX_VOICES; ++i) { v.age[i] = 0ull; v.steal_fade[i] = 0u; v.midi_of[i] = 0; }
    v.age_counter = 0ull; v.n_dev_ops = 0u; v.sustain_held = 0; v.mlp_enabled = 1; v.dev_overflow = 0; v.pad = 0;
}
OW_VM_HD int vm_ctz(uint64_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)m) - 1;
#else
    return __builtin_ctzll(m);
#endif
}
OW_VM_HD int vm_state_of(const OwVm& v, int s) {
    const uint64_t b = 1ull << s;
    return (v.st_mask[1] & b) ? 1 : (v.st_mask[2] & b) ? 2 : (v.st_mask[3] & b) ? 3 : 0;
}
OW_VM_HD void vm_set_state(OwVm& v, int s, int st) {
    const uint64_t b = 1ull << s;
    v.st_mask[0] &= ~b; v.st_mask[1] &= ~b; v.st_mask[2] &= ~b; v.st_mask[3] &= ~b;
    v.st_mask[st] |= b;
}
// bit s set <=> midi_of[s] == note (eight notes per 64-bit word: the zero-byte test of (word ^ pattern))
OW_VM_HD uint64_t vm_note_match(const OwVm& v, uint8_t note) {
    const uint64_t pat = 0x0101010101010101ull * (uint64_t)note;
    uint64_t m = 0;
    for (int k = 0; k < 8; ++k) {
        uint64_t w;
        __builtin_memcpy(&w, v.midi_of + 8 * k, 8);                       // little-endian: byte j of w = midi_of[8 k + j]
        const uint64_t x = w ^ pat;
        // exact per-byte zero test (no borrow across bytes): a byte of y has its top bit set <=> that byte of x is non-zero
        const uint64_t y = ((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x;
        const uint64_t z = ~y & 0x8080808080808080ull;                  // 0x80 in the bytes of x that are zero
        m |= (((z >> 7) * 0x0102040810204080ull) >> 56) << (8 * k);       // the eight flags gathered into one byte
    }
    return m;
}
OW_VM_HD void vm_sync_masks(OwVm& v, int s) {
    const uint64_t b = 1ull << s;
    if ((v.has_voice & b) && !(v.st_mask[OW_VOICE_FREE] & b)) v.main_mask |= b; else v.main_mask &= ~b;
    if (v.has_steal & b) v.steal_mask |= b; else v.steal_mask &= ~b;
}
OW_VM_HD int vm_allocate_voice(const OwVm& v) {  // engine.rs:569-590
    // first free slot, else the oldest voice of the lowest-priority non-empty class (releasing < sustained < held); ages are unique, so
    // this is the slot the reference's min_by_key over (class, age) returns
    if (v.st_mask[OW_VOICE_FREE]) return vm_ctz(v.st_mask[OW_VOICE_FREE]);
    const uint64_t cls = v.st_mask[OW_VOICE_RELEASING] ? v.st_mask[OW_VOICE_RELEASING]
                       : v.st_mask[OW_VOICE_SUSTAINED] ? v.st_mask[OW_VOICE_SUSTAINED] : v.st_mask[OW_VOICE_HELD];
    int best_idx = 0;
    uint64_t best = ~0ull;
    for (uint64_t m = cls; m; m &= m - 1) {
        const int i = vm_ctz(m);
        if (v.age[i] < best) { best = v.age[i]; best_idx = i; }
    }
    return best_idx;
}

// Sink: receives the slot ops of an event, `void push(uint8_t type, int slot, uint8_t note, bool mlp, uint32_t seed, double velocity)`.
template <typename Sink>
OW_VM_HD void vm_note_on(OwVm& v, Sink& sink, uint8_t note_in, float velocity, uint32_t fade) {  // engine.rs:299-338; fade = sr * 0.005 saturated to u32
    const uint8_t note = note_in < OW_MIDI_LO ? (uint8_t)OW_MIDI_LO : (note_in > OW_MIDI_HI ? (uint8_t)OW_MIDI_HI : note_in);
    if (v.st_mask[OW_VOICE_SUSTAINED]) {
        for (uint64_t m = v.st_mask[OW_VOICE_SUSTAINED] & vm_note_match(v, note); m; m &= m - 1) {
            const int i = vm_ctz(m);
            vm_set_state(v, i, OW_VOICE_RELEASING);
            if ((v.has_voice >> i) & 1ull) sink.push(OP_DAMPER, i, note, false, 0u, 0.0);
        }
    }
    const int idx = vm_allocate_voice(v);
    const uint64_t b = 1ull << idx;
    if (vm_state_of(v, idx) != OW_VOICE_FREE) {
        if (v.has_voice & b) {
            sink.push(OP_MOVE_STEAL, idx, note, false, fade, 0.0);
            v.has_steal |= b;
        } else {
            v.has_steal &= ~b;  // Option::take() of an empty voice
        }
        v.has_voice &= ~b;
        v.steal_fade[idx] = fade;
    }
    v.age_counter += 1;
    const uint32_t seed = (uint32_t)note * 2654435761u + (uint32_t)v.age_counter;
    sink.push(OP_NOTE_ON, idx, note, v.mlp_enabled != 0, seed, (double)velocity);
    v.has_voice |= b;
    vm_set_state(v, idx, OW_VOICE_HELD);
    v.midi_of[idx] = note;
    v.age[idx] = v.age_counter;
    vm_sync_masks(v, idx);
}
template <typename Sink>
OW_VM_HD void vm_note_off(OwVm& v, Sink& sink, uint8_t note_in) {  // engine.rs:340-359
    const uint8_t note = note_in < OW_MIDI_LO ? (uint8_t)OW_MIDI_LO : (note_in > OW_MIDI_HI ? (uint8_t)OW_MIDI_HI : note_in);
    if (!v.st_mask[OW_VOICE_HELD]) return;
    int oldest = -1;
    for (uint64_t m = v.st_mask[OW_VOICE_HELD] & vm_note_match(v, note); m; m &= m - 1) {
        const int i = vm_ctz(m);
        if (oldest < 0 || v.age[i] < v.age[oldest]) oldest = i;
    }
    if (oldest < 0) return;
    if (v.sustain_held) vm_set_state(v, oldest, OW_VOICE_SUSTAINED);
    else {
        vm_set_state(v, oldest, OW_VOICE_RELEASING);
        if ((v.has_voice >> oldest) & 1ull) sink.push(OP_DAMPER, oldest, note, false, 0u, 0.0);
    }
}
template <typename Sink>
OW_VM_HD void vm_set_sustain(OwVm& v, Sink& sink, bool held) {  // engine.rs:361-374
    if (v.sustain_held && !held) {
        for (uint64_t m = v.st_mask[OW_VOICE_SUSTAINED]; m; m &= m - 1) {   // ascending slot order, as the reference iterates
            const int i = vm_ctz(m);
            vm_set_state(v, i, OW_VOICE_RELEASING);
            if ((v.has_voice >> i) & 1ull) sink.push(OP_DAMPER, i, v.midi_of[i], false, 0u, 0.0);
        }
    }
    v.sustain_held = held ? 1 : 0;
}
