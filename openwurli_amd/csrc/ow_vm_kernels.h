// openwurli-hip: a burst of MIDI events applied on the device (ow_pool_midi on a big pool).  The state machine is ow_vm.h's -- the code the
// host runs -- one lane per engine over the engine's slice of the (engine-grouped) event list.
#pragma once
#include "ow_vm.h"
#include <hip/hip_runtime.h>

namespace owdev {

// begin[e] / end[e]: slice of engine e in the list (both 0 for an engine without events; the arrays are cleared before the launch)
// flags |= 2 when the list is not grouped by engine (an event whose engine index is below its predecessor's): the slices are then
// meaningless and the host drops the burst (it checks the word together with the queue-overflow bit)
__global__ void k_vm_index(const ow_midi_event* __restrict__ ev, size_t n, uint32_t* __restrict__ begin, uint32_t* __restrict__ end, uint32_t I,
                           uint32_t* __restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = ev[i].engine;
    if (i > 0 && ev[i - 1].engine > e) atomicOr(flags, 2u);
    if (e >= I) return;                                            // events for engines the pool does not have are ignored (as on the host)
    if (i == 0 || ev[i - 1].engine != e) begin[e] = (uint32_t)i;
    if (i + 1 == n || ev[i + 1].engine != e) end[e] = (uint32_t)(i + 1);
}

struct VmDevSink {       // the engine's op queue in HBM: OW_VM_OPS_MAX entries at a fixed place
    OwOp* base;
    uint32_t n;
    bool overflow;
    __device__ void push(uint8_t type, int slot, uint8_t note, bool mlp, uint32_t seed, double vel) {
        if (n >= OW_VM_OPS_MAX) { overflow = true; return; }
        OwOp op;
        op.type = type; op.slot = (uint8_t)slot; op.note = note; op.mlp = mlp ? 1 : 0; op.seed = seed; op.velocity = vel;
        base[n++] = op;
    }
};

// lane = engine: its events in list order (midi_apply_one of the host, openwurli_hip.hip)
__global__ __launch_bounds__(64) void k_vm_events(OwVm* __restrict__ vm, const ow_midi_event* __restrict__ ev, const uint32_t* __restrict__ begin,
                                                  const uint32_t* __restrict__ end, OwOp* __restrict__ ops_fix, uint32_t e_lo, uint32_t e_hi, uint32_t fade,
                                                  uint32_t* __restrict__ overflow) {
    const uint32_t e = e_lo + blockIdx.x * 64u + threadIdx.x;
    if (e >= e_hi) return;
    const uint32_t b = begin[e], f = end[e];
    if (b >= f) return;
    OwVm& v = vm[e];
    VmDevSink sink{ops_fix + (size_t)e * OW_VM_OPS_MAX, v.n_dev_ops, false};
    for (uint32_t i = b; i < f; ++i) {
        const ow_midi_event m = ev[i];
        if (m.type == 0) vm_note_on(v, sink, m.note, m.value, fade);
        else if (m.type == 1) vm_note_off(v, sink, m.note);
        else if (m.type == 2) vm_set_sustain(v, sink, m.value >= 0.5f);
    }
    v.n_dev_ops = sink.n;
    if (sink.overflow) { v.dev_overflow = 1; atomicOr(overflow, 1u); }
}

// after the queues of a burst have been applied right away (k_apply_ops launched by ow_pool_midi itself): they are empty again
__global__ void k_vm_clear_dev_ops(OwVm* __restrict__ vm, uint32_t e_lo, uint32_t e_hi) {
    const uint32_t e = e_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (e < e_hi) vm[e].n_dev_ops = 0u;
}

}  // namespace owdev
