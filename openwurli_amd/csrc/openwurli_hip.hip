// openwurli-hip: host side of the C-ABI (include/openwurli_hip.h) -- pool/engine objects,
// the voice-pool / MIDI state machine of WurliEngine (engine.rs:299-374,569-602) and the
// kernel launch sequence of one render.  The state machine is integer/branchy host work;
// everything that touches audio samples runs in the gfx950 kernels of ow_kernels.h.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <emmintrin.h>
#include <string>
#include <thread>
#include <vector>

#include "../../include/openwurli_hip.h"
#include "../../include/openwurli_hip_test.h"
#include "ow_consts_host.hpp"
#include "ow_vm.h"
#include "ow_kernels.h"
#include "ow_job_kernels.h"
#include "ow_mlp_mfma.h"
#include "ow_melange_dev.h"
#include "ow_melange_lit.h"
#include "ow_melange_col.h"
#include "ow_melange_eng.h"
#include "ow_power_amp_dev.h"
#include "ow_features.h"
#include "ow_trem_wide.h"
#include "ow_trem_row.h"
#include "ow_chain_row.h"
#include "ow_audit.h"
#include "ow_midi_kernels.h"
#include "ow_chain_wide.h"
#include "ow_chain_stream.h"
#include "ow_vm_kernels.h"
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>

using owdev::OwEngineOut;

namespace {

thread_local std::string g_err;
void set_err(const std::string& s) { g_err = s; }

// Master seed 0 of the reference = "entropy from the system clock" (gen_preamp.rs:1512-1521), taken once per process because
// every preamp clones one cached state (melange_adapter.rs:12-29).  OW_NOISE_SEED overrides it (reproducible runs).
uint64_t process_noise_seed() {
    static const uint64_t seed = []() -> uint64_t {
        if (const char* env = std::getenv("OW_NOISE_SEED")) { const unsigned long long v = std::strtoull(env, nullptr, 0); if (v) return (uint64_t)v; }
        const uint64_t t = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        return t ? t : (uint64_t)0x0123456789ABCDEFull;
    }();
    return seed;
}

#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Pinned host blocks handed out by ow_host_alloc: mapped into the device's address space, so the output stage of a big pool can store a
// rendered block straight into the caller's buffer (no d_out -> host copy trailing the last kernel).  render looks a target up here;
// anything else (pageable memory, blocks pinned by somebody else) takes the staged copy.
std::mutex g_host_mu;
struct HostBlock { size_t bytes; void* dptr; };
std::map<uintptr_t, HostBlock> g_host_blocks;
void host_block_register(void* ptr, size_t bytes, void* dptr) { std::lock_guard<std::mutex> lk(g_host_mu); g_host_blocks[(uintptr_t)ptr] = HostBlock{bytes, dptr}; }
void host_block_forget(void* ptr) { std::lock_guard<std::mutex> lk(g_host_mu); g_host_blocks.erase((uintptr_t)ptr); }
// device address of [ptr, ptr + bytes) when it lies inside one registered block, else nullptr
void* host_block_device_ptr(const void* ptr, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_host_mu);
    auto it = g_host_blocks.upper_bound((uintptr_t)ptr);
    if (it == g_host_blocks.begin()) return nullptr;
    --it;
    const uintptr_t off = (uintptr_t)ptr - it->first;
    if (!it->second.dptr || off > it->second.bytes || bytes > it->second.bytes - off) return nullptr;
    return (char*)it->second.dptr + off;
}

struct DevMem {   // device buffer released on every exit path
    void* p = nullptr;
    DevMem() = default;
    DevMem(const DevMem&) = delete;
    DevMem& operator=(const DevMem&) = delete;
    ~DevMem() { if (p) hipFree(p); }
    template <class T> T* as() const { return static_cast<T*>(p); }
    void alloc(size_t bytes) { HIP_OK(hipMalloc(&p, std::max<size_t>(bytes, 8))); }
};
struct StreamOwner {
    hipStream_t s = nullptr;
    ~StreamOwner() { if (s) hipStreamDestroy(s); }
};

// Persistent host worker threads.  The realtime entry points (ow_pool_render, ow_pool_midi) must not allocate once capacity is
// ensured (nih-plug's assert_process_allocs; engine.rs:288-297), and starting a std::thread allocates its state
// block and a stack: big pools therefore cut their per-engine host work into slices that run on these threads, started once
// per process and parked on a condition variable in between.  run() hands out slice indices from an atomic counter; the
// caller works too.  No std::function, no heap: the job is a function pointer + context pointer.
class Workers {
  public:
    static Workers& get() { static Workers w; return w; }
    // fn(ctx, t) for t in [0, T); returns when all slices are done.  One dispatch at a time (callers of different pools serialise).
    void run(size_t T, void (*fn)(void*, size_t), void* ctx) {
        // run() is not re-entered from inside a slice by anything in this library; if it ever is (in_slice_ set on this thread), the nested
        // call runs its slices inline -- std::mutex::try_lock on a mutex the calling thread already owns would be undefined behaviour.
        if (T <= 1 || th_.empty() || in_slice_) { for (size_t t = 0; t < T; ++t) fn(ctx, t); return; }
        // One dispatch at a time.  A second caller (another pool rendering on another audio thread) does NOT wait for the workers -- a
        // realtime thread blocked on a mutex that lower-priority work holds is a priority inversion -- it runs its slices itself, one
        // after the other: the same work as a single-threaded pass over the range (the slices partition it), i.e. the cost of a host with
        // one core, not a wait of unknown length.  Two pools of >= 16 384 engines rendering concurrently from two audio threads is the only
        // configuration that gets here.
        std::unique_lock<std::mutex> one(dispatch_mu_, std::try_to_lock);
        if (!one.owns_lock()) { for (size_t t = 0; t < T; ++t) fn(ctx, t); return; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = fn; ctx_ = ctx; total_ = T; next_.store(0); done_ = 0; ++gen_;
        }
        cv_.notify_all();
        size_t mine = 0;
        in_slice_ = true;
        for (size_t t; (t = next_.fetch_add(1)) < T; ++mine) fn(ctx, t);
        in_slice_ = false;
        std::unique_lock<std::mutex> lk(mu_);
        done_ += mine;
        cv_done_.wait(lk, [&] { return done_ == total_ && active_ == 0; });
        fn_ = nullptr;
    }
    template <class F> void each(size_t T, F& f) { run(T, [](void* c, size_t t) { (*static_cast<F*>(c))(t); }, &f); }
    size_t threads() const { return th_.size() + 1; }

  private:
    Workers() {
        size_t n = host_threads();
        for (size_t i = 1; i < n; ++i) th_.emplace_back([this] { loop(); });
    }
    ~Workers() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    static size_t host_threads();
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            void (*fn)(void*, size_t) = fn_;
            void* ctx = ctx_;
            const size_t T = total_;
            if (!fn) continue;
            ++active_;
            lk.unlock();
            size_t mine = 0;
            in_slice_ = true;
            for (size_t t; (t = next_.fetch_add(1)) < T; ++mine) fn(ctx, t);
            in_slice_ = false;
            lk.lock();
            done_ += mine;
            --active_;
            if (done_ == total_ && active_ == 0) cv_done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, dispatch_mu_;
    std::condition_variable cv_, cv_done_;
    void (*fn_)(void*, size_t) = nullptr;
    void* ctx_ = nullptr;
    size_t total_ = 0, done_ = 0, active_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;
    static thread_local bool in_slice_;
};
thread_local bool Workers::in_slice_ = false;
#define OW_MAX_STAGES 8
#define OW_MAX_SLICES 64   // upper bound of the slices one dispatch is cut into (scratch arrays live on the stack)

struct HostSmoother {  // host mirror of LinearSmoother::target only (the 1e-9 acceptance test, engine.rs:86-89)
    double target;
    bool pending = false;
    double pending_value = 0.0;
    void set_target(double t) {
        if (std::fabs(t - target) < 1e-9) return;
        target = t;
        pending = true;
        pending_value = t;
    }
};

// Measurement and test switches.  Every OW_* environment variable that selects between kernel paths is read ONCE, when a pool is
// created (std::getenv is neither realtime-safe nor safe against a concurrent setenv in the host), and kept in the pool; the render
// path only looks at the latched copy.  ow_test_pool_set_switch (openwurli_hip_test.h) changes one on a live pool.
struct Switches {
    int trem_wide = -1, preamp_wide = -1;      // -1: by pool size
    int chain_fused = -1;                      // OW_CHAIN_FUSED=0/1: preamp + output stage as one launch (k_chain_fused); -1: whenever the quad preamp is used
    bool trem_serial = false;                  // OW_TREM_SERIAL=1: block-ahead oscillators in front of the voices instead of beside them
    bool trem_cache = true;                    // OW_TREM_CACHE=0: no process-wide settled-state cache
    bool trem_traj = true;                     // OW_TREM_TRAJ=0: no shared trajectory, one oscillator per phase group (rounds 1-3)
    bool mel_rank1 = false, mel_lds = false, mel_generic = false;
    int mel_eng = 0;                           // OW_MEL_ENG=1: lane = engine melange kernel (k_preamp_mel_eng; measured -3 % at 131 072 engines, 2x slower at 65 536)
    bool voice_skew = true;                    // OW_VOICE_SKEW=0: the steady voice kernel without skewed lane clocks (one jitter grid per wavefront assumed)
    int pa_sort = 1;                           // OW_PA_SORT: 0 never, 1 when the block exceeds the chip, 2 always
    int eout_attn = -1;                        // OW_EOUT_ATTN=0/1: status summary instead of the status blocks (k_eout_attention); -1: ranges of >= 8 192 engines
    int pipe = 0;                              // OW_PIPE=n stages
    int midi_device = -1;                      // OW_MIDI_DEVICE=0/1: bursts of ow_pool_midi applied on the device (k_vm_events) never / whenever the list allows; -1: pools of >= 8 192 engines, >= 65 536 events
    int midi_apply_early = 1;                  // OW_MIDI_APPLY_EARLY=0: the queues of a device burst wait for the next render's k_apply_ops
    int voice_release = 1;                     // OW_VOICE_RELEASE=0: no release variant of the steady voice kernel (k_voice renders every engine with a damping voice)
    int voice_steal = 1;                       // OW_VOICE_STEAL=0: no steal variant of the steady voice kernel (k_voice renders every crossfade)
    int voice_attack = 1;                      // OW_VOICE_ATTACK=0: no attack variant of the steady voice kernel (engines in onset / noise phases go to the general kernel)
    bool force_general = false;                // test / probe hook: every engine's slot voices go to the general voice kernel (what it costs without any phase active)
    int chain_row = -1;                        // OW_CHAIN_ROW=0/1: the fused chain launch with one solver state per row of sixteen lanes (k_chain_row) never / whenever the chain is fused; -1: ranges of <= 1 024 engines
    int chain_stream = -1;                     // OW_CHAIN_STREAM=0/1: preamp + output stage of a big oversampled pool as one launch (k_chain_stream); -1: when the block goes to a pinned host block
    int out_direct = -1;                       // OW_OUT_DIRECT=0/1: output stage stores straight into a pinned host block (ow_host_alloc) instead of d_out + copy; -1: default
    bool pipe_overlap = false;
    bool host_profile = false;
    int midi_threads = 0;                      // OW_MIDI_THREADS
    static int flag(const char* name, int dflt) { const char* e = std::getenv(name); return (e && e[0]) ? (e[0] - '0') : dflt; }
    static Switches from_env() {
        Switches w;
        w.trem_wide = flag("OW_TREM_WIDE", -1); w.preamp_wide = flag("OW_PREAMP_WIDE", -1);
        if (w.trem_wide > 1 || w.trem_wide < -1) w.trem_wide = -1;
        if (w.preamp_wide > 1 || w.preamp_wide < -1) w.preamp_wide = -1;
        w.chain_fused = flag("OW_CHAIN_FUSED", -1);
        if (w.chain_fused > 1 || w.chain_fused < -1) w.chain_fused = -1;
        w.trem_serial = flag("OW_TREM_SERIAL", 0) == 1;
        w.trem_cache = flag("OW_TREM_CACHE", 1) != 0;
        w.trem_traj = flag("OW_TREM_TRAJ", 1) != 0;
        w.mel_rank1 = flag("OW_MEL_RANK1", 0) == 1; w.mel_lds = flag("OW_MEL_LDS", 0) == 1; w.mel_generic = flag("OW_MEL_GENERIC", 0) == 1;
        w.mel_eng = flag("OW_MEL_ENG", 0) == 1;
        w.eout_attn = flag("OW_EOUT_ATTN", -1); if (w.eout_attn > 1 || w.eout_attn < -1) w.eout_attn = -1;
        w.voice_skew = flag("OW_VOICE_SKEW", 1) != 0;
        w.pa_sort = flag("OW_PA_SORT", 1); if (w.pa_sort < 0 || w.pa_sort > 2) w.pa_sort = 1;
        if (const char* e = std::getenv("OW_PIPE")) { const int v = std::atoi(e); w.pipe = (v >= 1 && v <= 8) ? v : 0; }
        w.pipe_overlap = flag("OW_PIPE_OVERLAP", 0) == 1;
        w.voice_attack = flag("OW_VOICE_ATTACK", 1) != 0;
        w.voice_steal = flag("OW_VOICE_STEAL", 1) != 0;
        w.voice_release = flag("OW_VOICE_RELEASE", 1) != 0;
        w.midi_apply_early = flag("OW_MIDI_APPLY_EARLY", 1) != 0;
        w.midi_device = flag("OW_MIDI_DEVICE", -1); if (w.midi_device > 1 || w.midi_device < -1) w.midi_device = -1;
        w.chain_row = flag("OW_CHAIN_ROW", -1); if (w.chain_row > 1 || w.chain_row < -1) w.chain_row = -1;
        w.chain_stream = flag("OW_CHAIN_STREAM", -1); if (w.chain_stream > 1 || w.chain_stream < -1) w.chain_stream = -1;
        w.out_direct = flag("OW_OUT_DIRECT", -1); if (w.out_direct > 1 || w.out_direct < -1) w.out_direct = -1;
        w.host_profile = std::getenv("OW_HOST_PROFILE") != nullptr;
        if (const char* e = std::getenv("OW_MIDI_THREADS")) { const long v = std::atol(e); if (v >= 1 && v <= 64) w.midi_threads = (int)v; }
        return w;
    }
};

struct TremTraj;   // shared Twin-T / CdS trajectory of one (device, chain rate), below
std::atomic<uint64_t> g_ops_dropped{0};      // slot ops that found no room behind a device-side burst (render_range; OW_MIDI_APPLY_EARLY=0 only)

}  // namespace

struct ow_engine {
    ow_pool* pool = nullptr;
    size_t index = 0;
    bool owns_pool = false;
    // VoiceSlot[64] + pool counters (engine.rs:39-62): plain data shared with the device (ow_vm.h).  An engine of a pool points into the
    // pool's pinned array (a burst of events applied on the device is copied back into it wholesale); a device-less test engine owns one.
    OwVm* vm = nullptr;
    OwVm own_vm;
    ow_engine() { vm_init(own_vm); vm = &own_vm; }
    int state_of(int s) const { return vm_state_of(*vm, s); }
    void set_state(int s, int st) { vm_set_state(*vm, s, st); }
    bool rail_sag = true;           // melange power amp: rail sag (PowerAmp::new_at_sample_rate starts with it on, power_amp.rs:335-346)
    bool noise_on = false;          // melange preamp thermal noise (engine.rs:394-400); DkPreamp::new starts with off / 1.0
    double thermal_gain = 1.0;
    HostSmoother volume{0.5}, depth{0.5}, spk{0.0};
    uint64_t nan_guard_fires = 0, output_nan_resets = 0;
    std::vector<OwOp> ops;  // pending slot ops queued on the host, applied at the start of the next render (behind the device-queued ones)
    double sr = 0.0;            // host sample rate (steal crossfade length, engine.rs:318)
    uint8_t* dirty = nullptr;   // -> pool->dirty[index]: engine has pending ops / setter targets / changed masks
    uint8_t* dirty_any = nullptr;   // -> pool->dirty_any: some engine of the pool is dirty (lets a steady block skip the per-engine scans)
    uint32_t* host_ops_any = nullptr; // -> pool->host_ops_any: NUMBER of engines that hold host-queued ops (a burst then stays on the host: queue order)
    // (test before set: sixteen MIDI threads storing to the one shared byte on every event bounce its cache line -- 520 ms instead of 40 for
    // a 16.7 M-event re-strike; a read of an already-set flag stays shared)
    void mark() { if (dirty) { *dirty = 1; if (!__atomic_load_n(dirty_any, __ATOMIC_RELAXED)) __atomic_store_n(dirty_any, (uint8_t)1, __ATOMIC_RELAXED); } }
    void sync_masks(int s) { vm_sync_masks(*vm, s); mark(); }
    void touch() { mark(); }
    // Sink of the shared state machine on the host: the engine's own op list
    void push(uint8_t type, int slot, uint8_t note, bool mlp, uint32_t seed, double vel) {
        OwOp op;
        op.type = type; op.slot = (uint8_t)slot; op.note = note; op.mlp = mlp ? 1 : 0; op.seed = seed; op.velocity = vel;
        ops.push_back(op);
        if (host_ops_any && ops.size() == 1) __atomic_fetch_add(host_ops_any, 1u, __ATOMIC_RELAXED);     // once per engine and block, not per event
        mark();
    }
};

struct ow_pool {
    int device = 0;
    size_t I = 0;
    size_t Lcap = 0;
    OwConsts hc{};
    hipStream_t stream = nullptr;      // voices -> preamp -> output stage
    hipStream_t stream_trem = nullptr; // tremolo oscillator: no audio input (tremolo.rs:121), runs beside the voices
    hipEvent_t ev_trem[2] = {nullptr, nullptr};   // one per rbuf half
    // The tremolo oscillator is produced one block ahead (speculating that the next block has the same length); the
    // tremolo rows of the chain state are backed up first so a mis-speculation can be rolled back.
    double* d_trem_backup = nullptr;   // [18][I]
    int rb_cur = 0;                    // rbuf half holding the R samples of the block being rendered
    struct { bool valid = false; int e0 = 0, ne = 0, n_os = 0; } spec;
    OwConsts* dK = nullptr;     // constants at the pool's rates
    OwConsts* dK48 = nullptr;   // tremolo codegen-rate matrices for CircuitState::warmup
    double* d_nt = nullptr;
    double* d_vrec = nullptr;
    double* d_cs = nullptr;
    int power_amp_kind = 0;           // OW_POWER_AMP_BEHAVIORAL / OW_POWER_AMP_MELANGE
    int tremolo_kind = 0;             // OW_TREMOLO_TWIN_T / OW_TREMOLO_LEGACY_LFO (the reference's `legacy-tremolo` cargo feature)
    OwPaConsts* dPa = nullptr;        // melange power amp: constants at the chain rate
    double* d_pa = nullptr;           // melange power amp: per-engine state, [PAS_COUNT][I]
    double* d_pa_settled = nullptr;   // settled circuit state (PAS_CIRCUIT_END doubles), power_amp.rs:288-296
    double* d_pa_tap = nullptr;       // test tap: amp output per chain-rate sample, [2 * Lcap][I] (ow_test_pool_enable_power_amp_tap)
    uint32_t* d_pa_demand = nullptr;  // [I] Newton passes of the engine's last block (k_post_mpa), 0 = not rendered yet
    uint32_t* d_pa_order = nullptr;   // [I] engines of a launch range by falling demand (k_pa_order_*)
    uint32_t* d_pa_hist = nullptr;    // [OW_MAX_STAGES][256] class counts / cursors of the ranges
    size_t pa_tap_cap = 0;
    double* d_mel_settled = nullptr;  // melange preamp: settled codegen-rate state (18 doubles)
    size_t mel_lu_ld = 0;             // column-streamed literal kernel: lanes per row of d_mel_lu
    double* d_mel_lu = nullptr;       // literal kernel: LU workspace of the generic rebuild, [ceil(I/32) + OW_MAX_SLICES + 1][12][12][32]
    double* d_noise = nullptr;        // melange preamp: thermal-noise state of the main solver states, [NZ_COUNT][I]
    double* d_sum = nullptr;
    double* d_rbuf = nullptr;
    double* d_pre = nullptr;
    float* d_out = nullptr;
    size_t out_ld = 0;                // row stride of d_out for the block it holds: rows are packed at the block length, so that the copy of
                                      // a block to the host is ONE linear transfer (copy engine) instead of a pitched one (blit kernel)
    OwEngineArgs* d_args = nullptr;
    OwEngineOut* d_eout = nullptr;
    OwOp* d_ops = nullptr;
    size_t ops_cap = 0;
    OwEngineArgs* h_args = nullptr;   // pinned
    OwEngineOut* h_eout = nullptr;    // pinned
    // Status summary of a block (big pools): k_eout_attention marks the engines whose status block the host has to look at (a voice
    // fell silent, a steal fade is running, a guard fired, the transient flag changed); the host copies one bit per engine and fetches
    // the status blocks themselves only when a bit is set -- a steady block of 131 072 engines then costs the host 16 KB instead of a
    // 5 MB copy and a 131 072-entry scan.
    uint32_t* d_skew_seen = nullptr;  // k_voice_steady: some wavefront of the launch held voices on more than one 16-sample jitter grid
    uint32_t* h_skew_seen = nullptr;  // pinned
    bool skew_next = false, skew_pending = false;   // variant of the next steady launch; a report is on its way
    uint64_t* d_attn = nullptr;       // [ceil(I / 64)]
    uint64_t* h_attn = nullptr;       // pinned
    uint8_t* d_prev_tr = nullptr;     // [I] transient flag the host knows (p->transient)
    uint8_t* h_prev_tr = nullptr;     // pinned staging of p->transient for a resync
    bool attn_pending = false;        // the block just rendered left its summary in h_attn instead of its status blocks in h_eout
    bool attn_resync = true;          // d_prev_tr has to be refreshed from p->transient before the next summary
    bool eout_all_live = true;        // a block went through the status-block path: any h_eout entry may hold something
    std::vector<uint32_t> eout_live;  // engines whose h_eout entry holds something other than "nothing happened" (cleared next block)
    uint8_t dirty_any = 1;            // some dirty[] entry may be set (engines set it; a whole-pool render clears it)
    bool any_cache_valid = false, any_main_c = false, any_steal_c = false;   // any_main / any_steal of the last whole-pool render
    OwEngineOut* d_eout_packed = nullptr;   // [I] status blocks of a list of engines, packed (voice-sum NaN guard's second pass)
    OwEngineOut* h_eout_packed = nullptr;   // pinned
    OwOp* h_ops = nullptr;            // pinned
    // Voice-pool states (ow_vm.h): h_vm is what the engines' host state machine works on; a burst of events (ow_pool_midi on a big pool) is
    // applied to d_vm by k_vm_events and copied back.  vm_host_dirty: the host changed some state since d_vm was last written.
    OwVm* h_vm = nullptr;             // pinned [I]
    OwVm* d_vm = nullptr;             // [I], allocated with the first burst
    OwOp* d_ops_fix = nullptr;        // [I][OW_VM_OPS_MAX] op queues written by the device
    ow_midi_event* h_ev = nullptr;    // pinned staging of a burst's events (lists that are not in a pinned block themselves)
    ow_midi_event* d_ev = nullptr;
    size_t ev_cap = 0;
    uint32_t* d_ev_begin = nullptr;   // [2][I] slice of every engine in the burst's list
    uint8_t vm_host_dirty = 1;
    uint32_t host_ops_any = 0;        // engines that hold host-queued ops (counted where a queue becomes non-empty / is drained: a host that mixes
                                      // single-engine events with partial renders gets the device bursts back as soon as the queues are empty)
    uint32_t* d_vm_ovf = nullptr;     // a device queue overflowed during the burst (the burst is then replayed on the host)
    uint32_t* h_vm_ovf = nullptr;     // pinned
    bool vm_download_pending = false; // d_vm -> h_vm is in flight (ev_vm)
    hipEvent_t ev_vm = nullptr;
    hipEvent_t ev_vm_events = nullptr;   // k_vm_events of the last burst has finished (the download waits for it on its own stream)
    // Once a pool has taken a burst on the device, the host's copy of the states goes up again in the BACKGROUND whenever a block's
    // book-keeping has changed it (end of ow_pool_render, copy stream, beside the next block's kernels): the next burst then finds the
    // device's copy current instead of starting with 126 MB of upload.  A host change while that copy is in flight sets vm_host_dirty
    // again (vm_host_changed comes first), and the burst uploads as before.
    hipEvent_t ev_vm_up = nullptr;
    bool vm_upload_inflight = false;
    // The last burst's queues were applied at once (k_apply_ops launched by ow_pool_midi, beside the download of the states): the engines
    // of [applied_lo, applied_hi) whose downloaded state still says "n_dev_ops queued" have nothing queued any more -- vm_wait_download
    // settles that on the host's copy and marks them as inside a note-on's phases for the next block's dispatch.
    bool dev_ops_applied = false;
    uint32_t applied_lo = 0, applied_hi = 0;
    bool dev_ops_pending = false;     // some engine's next ops sit in d_ops_fix
    uint64_t vm_bursts = 0;           // bursts that went through the device (test hook)
    struct OpTail { uint32_t src, dst, n; };   // host-queued ops of an engine that also has device-queued ones: appended on the device
    std::vector<OpTail> op_tails;
    std::mutex op_tails_mu;
    // Packed voice dispatch (ow_kernels.h): lane = sounding voice.  Three block lists of (engine << 6 | slot) entries, rebuilt when a
    // mask, a pending op or a transient flag changed: steady slot voices, slot voices of engines in a transient phase, steal voices.
    // sig: signature of the (engine, mask) sequence the device copy was packed from (two independent 64-bit hashes + layout); a rebuild that
    // arrives at the same signature leaves the list alone (the blocks after a whole-pool re-strike repack 33 MB per list otherwise)
    struct VoiceList { uint32_t* h = nullptr; uint32_t* d = nullptr; uint32_t n_blocks = 0; uint64_t sig[3] = {0, 0, 0}; bool sig_valid = false; };
    VoiceList vl_steady, vl_general, vl_steal, vl_attack;     // vl_attack: engines inside onset ramps / attack noise whose slot voices are not damping (k_voice_steady<false, 1>)
    std::vector<uint8_t> transient;   // per engine: device status after the previous block (OwEngineOut::transient)
    bool lists_valid = false;
    int lists_e0 = -1, lists_ne = -1;
    // Staged render (render_range), used when a big pool's block is copied to the host: the range is cut into NP engine stages, each
    // on its own stream; the kernels of stage k start when those of stage k-1 have finished, so the device-to-host copy of stage k-1
    // (copy engine) runs beside the kernels of stage k.  Stage boundaries are slice boundaries of the packed voice lists.
    int slice_T = 1, slice_per = 0;                             // slices the lists were packed in, engines per slice
    struct SliceStart { uint32_t s = 0, g = 0, t = 0, a = 0; uint64_t h[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}; } slice_start[OW_MAX_SLICES + 1];   // entry offsets of every slice in the four lists (h: pass-1 hashes of a slice, unused in the stored copy)
    hipStream_t pipe_stream[OW_MAX_STAGES] = {};               // [0] == stream
    hipEvent_t ev_ready = nullptr, ev_voice_done[OW_MAX_STAGES] = {}, ev_stage_done[OW_MAX_STAGES] = {};
    hipEvent_t ev_stage[OW_MAX_STAGES][5] = {};                // profiling: before voices, after voices, before preamp, after preamp, after post
    int last_np = 1;
    uint32_t* d_op_engines = nullptr; // engines that have pending ops this block (k_apply_ops runs one block per entry)
    uint32_t* h_op_engines = nullptr; // pinned, I entries
    std::vector<ow_engine*> engines;
    std::vector<uint8_t> dirty;       // per engine: host state changed since the args were last uploaded
    bool args_stale = true;           // device args still hold one-shot fields of the previous block
    // Tremolo phase groups: engines whose Twin-T / CdS state is bit-identical (everything since their last chain init happened in
    // lock-step) share ONE oscillator: the group's leader (its lowest engine index) carries the 18 tremolo rows of the chain state and
    // owns a column of rbuf, the other members read that column.  A fresh pool is one group; ow_engine_reset / ow_engine_warm_up of a
    // single engine split it off (per-engine fallback), a whole-pool reset / rate change merges everything again.
    uint32_t* h_lead = nullptr;       // pinned [I]: leader engine of every engine
    uint32_t* d_lead = nullptr;
    uint32_t* h_leaders = nullptr;    // pinned [I]: compact list of the leaders inside the range being rendered
    uint32_t* d_leaders = nullptr;
    uint32_t* h_copy = nullptr;       // pinned [2][I]: (src, dst) pairs for k_trem_copy_rows
    uint32_t* d_copy = nullptr;
    std::vector<uint32_t> grp_in, grp_out;   // scratch of trem_split_at_range (first member inside / outside the range, per leader)
    int n_lead = 0, lead_e0 = -1, lead_ne = -1;
    bool lead_list_valid = false;     // d_leaders matches (lead_e0, lead_ne) and the current groups
    int split_e0 = -1, split_ne = -1; // range for which "no group straddles the range boundary" is known to hold
    double* d_snap = nullptr;         // [3][I] smoother targets (depth, speaker, volume) handed to k_chain_init on reset
    double* h_snap = nullptr;         // pinned
    double* d_trem_settled = nullptr; // staging of one cached settled Twin-T state (18 doubles), see trem_settled_rows
    uint32_t* d_zero = nullptr;       // one zero (leader list {0} of the I = 1 settle scratch)
    // Shared trajectory (TremTraj): engines of a Twin-T pool read r_ldr[t] of ONE process-wide sequence at t = trem_clock - birth[e].
    // trem_clock advances with every whole-pool block; a sub-range rendered on its own shifts the births of its engines instead.
    // birth == OW_OFF_TRAJ: the engine left the trajectory (older than the store's cap) and owns a phase group of one.
    std::shared_ptr<TremTraj> traj;
    long long trem_clock = 0;
    long long min_birth = 0;          // over the engines on the trajectory (the oldest one decides how far the store must reach)
    size_t n_on_traj = 0;
    std::vector<long long> h_birth;   // [I]
    long long* d_birth = nullptr;
    unsigned long long* d_evict = nullptr;   // [3][I] scratch of trem_evict (engine list, positions, fallback counts): render never allocates
    long long last_n_os = 0;          // chain-rate samples of the last rendered block (ow_pool_read_tremolo_r)
    Switches sw;                      // latched at creation
    bool voices_only = false;         // ow_render_note: the pool renders voice sums only -- no chain state, no tremolo / preamp / output kernels
    int inject_faults = 0;            // test hook (openwurli_hip_test.h): the next n renders fail before their first launch
    bool steal_counted = false;       // the block in flight had its steal fades counted down while its kernels ran (steal_countdown_early)
    double hostprof_acc[4] = {0, 0, 0, 0};
    long hostprof_cnt = 0;
    bool profiling = false;
    hipEvent_t ev[8] = {};
    float last_ms[5] = {0, 0, 0, 0, 0};
    size_t last_len = 0;
};

namespace {

// The host mirror of the voice-pool states is current (a burst applied on the device is copied back asynchronously: wait for it), and
// the host is about to change / has changed one of them (the device copy is stale until the next burst uploads them again).
void vm_settle_applied(ow_pool* p);
void vm_wait_download(ow_pool* p) {
    if (p && p->vm_download_pending) { hipEventSynchronize(p->ev_vm); p->vm_download_pending = false; if (p->dev_ops_applied) vm_settle_applied(p); }
}
void vm_host_current(const ow_engine* e) { if (e && e->pool) vm_wait_download(e->pool); }
void vm_host_changed(ow_engine* e) { if (e && e->pool && !__atomic_load_n(&e->pool->vm_host_dirty, __ATOMIC_RELAXED)) __atomic_store_n(&e->pool->vm_host_dirty, (uint8_t)1, __ATOMIC_RELAXED); }

// Drop a pending block-ahead tremolo result: restore the oscillator rows it advanced and drain the tremolo stream.
void invalidate_spec(ow_pool* p) {
    if (p->spec.valid) {
        hipMemcpyAsync(p->d_cs, p->d_trem_backup, sizeof(double) * 18 * p->I, hipMemcpyDeviceToDevice, p->stream_trem);
        p->spec.valid = false;
    }
    if (p->stream_trem) hipStreamSynchronize(p->stream_trem);
}

void free_stream_buffers(ow_pool* p) {
    invalidate_spec(p);
    if (p->d_sum) hipFree(p->d_sum);
    if (p->d_rbuf) hipFree(p->d_rbuf);
    if (p->d_pre) hipFree(p->d_pre);
    if (p->d_out) hipFree(p->d_out);
    p->d_sum = p->d_rbuf = p->d_pre = nullptr;
    p->d_out = nullptr;
}

void alloc_stream_buffers(ow_pool* p, size_t cap) {
    free_stream_buffers(p);
    const size_t I = p->I;
    HIP_OK(hipMalloc(&p->d_sum, sizeof(double) * 2 * I * cap));
    HIP_OK(hipMalloc(&p->d_rbuf, sizeof(double) * 2 * (2 * cap * I)));   // two halves (current block, block ahead)
    HIP_OK(hipMalloc(&p->d_pre, sizeof(double) * 2 * cap * I));
    HIP_OK(hipMalloc(&p->d_out, sizeof(float) * I * cap));
    HIP_OK(hipMemsetAsync(p->d_out, 0, sizeof(float) * I * cap, p->stream));
    if (p->d_pa_tap) {      // the test tap follows the block capacity
        hipFree(p->d_pa_tap); p->d_pa_tap = nullptr;
        HIP_OK(hipMalloc(&p->d_pa_tap, sizeof(double) * 2 * cap * I));
    }
    p->Lcap = cap;
}

// Op staging (pinned + device).  pool_create sizes it for a whole-keyboard re-strike of EVERY engine in one block (damper +
// move-to-steal + note-on per key = 192 ops of 16 B per engine), so ow_pool_render never gets here with n > ops_cap unless more
// than that was queued between two renders -- the only case in which a render allocates (like the reference's buffer auto-grow).
void ensure_ops_capacity(ow_pool* p, size_t n) {
    if (n <= p->ops_cap) return;
    size_t cap = std::max<size_t>(n, std::max<size_t>(p->ops_cap * 2, 256));
    if (p->d_ops) hipFree(p->d_ops);
    if (p->h_ops) hipHostFree(p->h_ops);
    p->d_ops = nullptr; p->h_ops = nullptr; p->ops_cap = 0;   // a failed allocation below must not leave freed pointers behind
    HIP_OK(hipMalloc(&p->d_ops, sizeof(OwOp) * cap));
    HIP_OK(hipHostMalloc(&p->h_ops, sizeof(OwOp) * cap));
    p->ops_cap = cap;
}

// Settled state of the melange preamp (melange_adapter.rs:12-20): rate-independent (always computed at the 48 kHz codegen
// tables), so it is produced once per device by k_mel_settle and cached, like the reference's OnceLock.
std::mutex g_mel_mu;
std::map<int, std::vector<double>> g_mel_settled;
void mel_settled_to_device(int device, double* d_dst, hipStream_t st) {
    // the process-wide mutex guards the maps only: a cached state is copied out under it and transferred after it is released (a pool
    // with a long queued stream must not hold up every other thread's pool creation / reset)
    std::vector<double> h;
    { std::lock_guard<std::mutex> lk(g_mel_mu); auto it = g_mel_settled.find(device); if (it != g_mel_settled.end()) h = it->second; }
    if (h.empty()) {
        OwConsts k48;
        owhip::build_consts(k48, 24000.0, OW_PREAMP_MELANGE12);   // os_sr = 48 kHz -> codegen tables
        DevMem dk;
        dk.alloc(sizeof(OwConsts));
        HIP_OK(hipMemcpyAsync(dk.p, &k48, sizeof(OwConsts), hipMemcpyHostToDevice, st));
        owdev::k_mel_settle<<<dim3(1), dim3(64), 0, st>>>(dk.as<OwConsts>(), d_dst);
        h.resize(18);
        HIP_OK(hipMemcpyAsync(h.data(), d_dst, sizeof(double) * 18, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        std::lock_guard<std::mutex> lk(g_mel_mu);
        g_mel_settled[device] = h;          // two threads racing here computed the same bits
        return;
    }
    HIP_OK(hipMemcpyAsync(d_dst, h.data(), sizeof(double) * 18, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
}

// Settled state of the melange power amp (power_amp.rs:288-296): always computed with the codegen-rate matrices, once per device.
std::map<int, std::vector<double>> g_pa_settled;
void pa_settled_to_device(int device, double* d_dst, hipStream_t st) {
    std::vector<double> h;
    { std::lock_guard<std::mutex> lk(g_mel_mu); auto it = g_pa_settled.find(device); if (it != g_pa_settled.end()) h = it->second; }
    if (h.empty()) {
        std::unique_ptr<OwPaConsts> h88(new OwPaConsts());
        owhip::build_pa_consts(*h88, PA_SAMPLE_RATE);
        DevMem dk;
        dk.alloc(sizeof(OwPaConsts));
        HIP_OK(hipMemcpyAsync(dk.p, h88.get(), sizeof(OwPaConsts), hipMemcpyHostToDevice, st));
        owdev::k_mpa_settle<<<dim3(1), dim3(64), 0, st>>>(dk.as<OwPaConsts>(), d_dst);
        HIP_OK(hipGetLastError());
        h.resize(owdev::PAS_CIRCUIT_END);
        HIP_OK(hipMemcpyAsync(h.data(), d_dst, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        std::lock_guard<std::mutex> lk(g_mel_mu);
        g_pa_settled[device] = h;
        return;
    }
    HIP_OK(hipMemcpyAsync(d_dst, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
}

// Settled state of the Twin-T oscillator (Tremolo::new: CircuitState::default's 50 warm-up steps at the codegen matrices, then 2 s of
// oscillator steps at the chain rate, tremolo.rs:92-102).  It depends on nothing but the chain rate, and it is 192 050 strictly serial
// solver steps -- 0.9 s on one wavefront -- which every pool creation, reset and rate change used to pay.  The reference caches its
// expensive settles the same way (OnceLock: melange_adapter.rs:12-29, power_amp.rs:283-299).  Keyed by (device, chain rate); the cached
// rows are what the product kernels produced on the first use, so a hit is bit-identical to a fresh settle
// (tests/test_gpu_boundary.py::test_settled_tremolo_cache_is_bit_identical).  OW_TREM_CACHE=0 disables it.
// OW_TREM_ROW=0: the quad-lane oscillator step (ow_trem_wide.h) for the settle and the shared trajectory instead of the row step
// (ow_trem_row.h).  Same samples either way (tests/test_gpu_trajectory.py); read once per process.
bool trem_row_enabled() { static const bool on = Switches::flag("OW_TREM_ROW", 1) != 0; return on; }
struct TremSettled { double rows[18]; };       // rows 0..16 after the settle; [17] = BE fallbacks the settle itself counted (u64 bits)
std::map<std::pair<int, uint64_t>, TremSettled> g_trem_settled;

// Settled rows for (device, chain rate), from the cache or by running the settle on `st` with the product kernels (k_tremolo_wide<true>:
// 50 steps at the codegen matrices, 2 s at the chain rate) in the 18-row scratch `d_state` (I = 1 layout).  use_cache = false: always
// settle, never store (OW_TREM_CACHE=0).  On return d_state holds the settled rows with row 17 = the settle's own fallback count.
TremSettled trem_settled_rows(int device, double os_sr, const OwConsts* dK, const OwConsts* dK48, double* d_state, const uint32_t* d_zero, hipStream_t st,
                              bool use_cache) {
    uint64_t rate_bits; std::memcpy(&rate_bits, &os_sr, 8);
    const std::pair<int, uint64_t> key(device, rate_bits);
    TremSettled t;
    if (use_cache) {
        bool hit = false;
        { std::lock_guard<std::mutex> lk(g_mel_mu); auto it = g_trem_settled.find(key); if (it != g_trem_settled.end()) { t = it->second; hit = true; } }
        if (hit) {     // copied out under the lock; the transfer and its sync run without it
            HIP_OK(hipMemcpyAsync(d_state, t.rows, sizeof(double) * 18, hipMemcpyHostToDevice, st));
            HIP_OK(hipStreamSynchronize(st));
            return t;
        }
    }
    const long long n_settle = (long long)owhip::sat_u32(os_sr * 2.0);
    owdev::k_trem_state_dc<<<dim3(1), dim3(64), 0, st>>>(d_state);
    if (trem_row_enabled()) {      // one system per wavefront (ow_trem_row.h): bit-identical, ~0.55 x the time
        owdev::k_trem_settle_row<<<dim3(1), dim3(64), 0, st>>>(dK48, d_state, 50LL);
        owdev::k_trem_settle_row<<<dim3(1), dim3(64), 0, st>>>(dK, d_state, n_settle);
    } else {
        owdev::k_tremolo_wide<true><<<dim3(1), dim3(64), 0, st>>>(dK48, d_state, nullptr, 1, 50LL, d_zero, 1);
        owdev::k_tremolo_wide<true><<<dim3(1), dim3(64), 0, st>>>(dK, d_state, nullptr, 1, n_settle, d_zero, 1);
    }
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(t.rows, d_state, sizeof(double) * 18, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (use_cache) { std::lock_guard<std::mutex> lk(g_mel_mu); g_trem_settled[key] = t; }
    return t;
}

// ---- shared Twin-T / CdS trajectory --------------------------------------------------------------------------------------------------
// Tremolo::process reads no audio and no depth (tremolo.rs:121-146) and every Tremolo::new / reset leaves the same settled state
// (:83-102, :192-216): r_ldr[t], t = process() calls since the cell was built, is one deterministic sequence per chain rate.  The store
// holds it in HBM once per (device, chain rate) for the life of the process; the engine with the largest t extends it with ONE oscillator
// (k_trem_traj_extend, a single wavefront on the store's own stream), every other engine of every pool reads it.  288 GB of HBM is what
// makes this the natural layout: 0.77 MB per second of audio at 96 kHz, OW_TREM_TRAJ_SECONDS (default 1800) of it reserved up front so
// that render never allocates.  Engines that grow older than that leave the store for an oscillator of their own (trem_evict).
struct TremTraj {
    int device = 0;
    double os_sr = 0.0;
    std::mutex mu;                    // guards len / cap / buffers / marks / enqueues on `stream`; held for host-side enqueue work only
    OwConsts* dK = nullptr;           // constants at the chain rate (the tremolo fields are all the kernels read)
    double* d_r = nullptr;            // [cap + 64]
    double* d_state = nullptr;        // [18] oscillator rows at sample `len`
    double* d_ckpt = nullptr;         // [cap / OW_TRAJ_CK + 2][OW_TRAJ_CKD]
    unsigned long long* d_be = nullptr;   // [1 + OW_TRAJ_BE_CAP]
    uint32_t* d_zero = nullptr;       // leaders = {0} for the settle kernels
    size_t cap = 0, len = 0;          // cap: samples the buffers hold; len: samples produced or enqueued for production
    size_t cap_max = 0;               // configured capacity (ow_tremolo_configure / OW_TREM_TRAJ_SECONDS): the buffers grow up to it, an engine
                                      // older than this leaves the store (trem_evict)
    size_t lead = 0;                  // samples the store is kept ahead of a fast reader (a small pool) by the feeder thread
    size_t target = 0;                // where the feeder is taking the store (<= cap)
    size_t reader_end = 0, reader_block = 1024;   // furthest sample a fast reader asked for, and its block (feeder back-off)
    std::chrono::steady_clock::time_point reader_seen{};
    size_t done = 0;                  // samples known to be complete (a recorded mark was seen finished)
    uint64_t be_settle = 0;           // fallbacks the settle itself counted (what a fresh CircuitState carries after Tremolo::new)
    hipStream_t stream = nullptr;
    static constexpr int NMARK = 16;
    struct Mark { size_t end = 0; hipEvent_t ev = nullptr; } mark[NMARK];
    int head = 0;
    bool grow_requested = false;      // the helper thread has been asked to double the buffers
    std::vector<void*> retired;       // buffers a growth replaced: kernels launched before the swap may still read them, so they are only
                                      // freed at the NEXT growth (minutes of audio later) or with the store
    ~TremTraj() {
        hipSetDevice(device);
        if (stream) hipStreamSynchronize(stream);
        for (auto& m : mark) if (m.ev) hipEventDestroy(m.ev);
        if (dK) hipFree(dK); if (d_r) hipFree(d_r); if (d_state) hipFree(d_state); if (d_ckpt) hipFree(d_ckpt);
        for (void* q : retired) hipFree(q);
        if (d_be) hipFree(d_be); if (d_zero) hipFree(d_zero);
        if (stream) hipStreamDestroy(stream);
    }
    static size_t ckpt_doubles(size_t c) { return (size_t)OW_TRAJ_CKD * (c / OW_TRAJ_CK + 2); }
    // Buffers for `new_cap` samples: allocate (no lock held: hipMalloc may take milliseconds), then under the lock copy what exists on the
    // store's stream -- behind every extension already enqueued -- and swap.  Readers that fetch the new pointer wait for the copy's mark.
    void grow_to(size_t new_cap) {
        new_cap = std::min((new_cap + OW_TRAJ_CK - 1) / OW_TRAJ_CK * OW_TRAJ_CK, cap_max);
        { std::lock_guard<std::mutex> lk(mu); if (new_cap <= cap) { grow_requested = false; return; } }
        HIP_OK(hipSetDevice(device));
        DevMem nr, nc;                                     // released on every exit path until the swap below takes them over
        nr.alloc(sizeof(double) * (new_cap + 64));
        nc.alloc(sizeof(double) * ckpt_doubles(new_cap));
        std::vector<void*> to_free;                        // hipFree drains the whole device: never under `mu` (a render thread in cover() would wait for it)
        {
            std::lock_guard<std::mutex> lk(mu);
            if (new_cap <= cap) { grow_requested = false; return; }
            to_free.swap(retired);                         // replaced one growth ago
            HIP_OK(hipMemcpyAsync(nr.p, d_r, sizeof(double) * (len + 64 <= cap + 64 ? len + 64 : cap + 64), hipMemcpyDeviceToDevice, stream));
            HIP_OK(hipMemcpyAsync(nc.p, d_ckpt, sizeof(double) * ckpt_doubles(cap), hipMemcpyDeviceToDevice, stream));
            Mark& m = mark[head];
            head = (head + 1) % NMARK;
            HIP_OK(hipEventRecord(m.ev, stream));
            m.end = len;
            retired.push_back(d_r); retired.push_back(d_ckpt);
            d_r = nr.as<double>(); d_ckpt = nc.as<double>(); cap = new_cap;
            nr.p = nullptr; nc.p = nullptr;
            done = 0;                                      // everything has to be waited for again (the copy)
            for (Mark& o : mark) if (&o != &m && o.end <= len) o.end = 0;     // older marks stand for data in the old buffer
            grow_requested = false;
        }
        for (void* q : to_free) hipFree(q);
    }
    // callers hold mu
    void extend_to(size_t end) {
        end = std::min(end, cap);
        if (end <= len) return;
        if (trem_row_enabled()) owdev::k_trem_traj_extend_row<<<dim3(1), dim3(64), 0, stream>>>(dK, d_state, d_r + len, (long long)len, (long long)(end - len), d_ckpt, d_be);
        else owdev::k_trem_traj_extend<<<dim3(1), dim3(64), 0, stream>>>(dK, d_state, d_r + len, (long long)len, (long long)(end - len), d_ckpt, d_be);
        HIP_OK(hipGetLastError());
        Mark& m = mark[head];
        head = (head + 1) % NMARK;
        HIP_OK(hipEventRecord(m.ev, stream));
        m.end = end;
        len = end;
    }
    // extension launches not known to have finished (their marks)
    int in_flight() {
        int n = 0;
        for (Mark& m : mark) if (m.end > done) { if (hipEventQuery(m.ev) == hipSuccess) done = std::max(done, m.end); else ++n; }
        return n;
    }
    // Make samples [0, end) exist (enqueue what is missing on the store's stream), then keep the store `ahead` samples further (one block:
    // what the next render will ask for).  lead_for > 0: the caller is a reader that can outrun real time (a small pool): the store's
    // target moves to end + lead_for and the feeder thread (TrajGrower) keeps the oscillator running towards it -- two short launches in
    // flight at a time, whether or not anybody renders meanwhile.  (Deep queues do not work here: a consumer stream that waits for a mark
    // of this stream was measured to resume only when everything enqueued on it had finished -- with a second of oscillator steps queued,
    // every block of a pool at the frontier waited 0.4 s.)  Returns true in *feed when the feeder has something new to do.
    // Returns the event a consumer stream has to wait for before it reads [0, end), or nullptr when they are known to be complete.  (The
    // event may be re-recorded by a later extension before the consumer waits on it: it then stands for a longer prefix -- still sufficient.)
    hipEvent_t cover(size_t end, size_t ahead, size_t lead_for = 0, bool* feed = nullptr) {
        end = std::min(end, cap);
        extend_to(end);
        hipEvent_t wait = nullptr;
        if (end > done) {
            const Mark* best = nullptr;
            for (const Mark& m : mark) if (m.end >= end && (!best || m.end < best->end)) best = &m;
            if (best) {
                if (hipEventQuery(best->ev) == hipSuccess) done = std::max(done, best->end);
                else wait = best->ev;
            }
        }
        if (ahead) extend_to(end + ahead);
        if (lead_for) {
            const size_t t = std::min(end + lead_for, cap);
            if (t > target) { target = t; if (feed && len < target) *feed = true; }
            reader_end = std::max(reader_end, end);
            if (ahead) reader_block = ahead;
            reader_seen = std::chrono::steady_clock::now();
        }
        return wait;
    }
    // Feeder thread: one more short launch towards `target` when fewer than two are in flight; true while there is more to do.
    // It stands back while a reader renders right behind the frontier (a pool that outruns the oscillator): such a reader enqueues exactly
    // the block ahead it needs (`ahead` above), and anything the feeder put in front of that would only be more for it to wait for.  The
    // lead is built whenever readers are further back or idle -- between a host's instantiation and its first block, between blocks of
    // a host paced at real time.
    bool feed_step() {
        std::lock_guard<std::mutex> lk(mu);
        const size_t t = std::min(target, cap);
        if (len >= t) return false;
        const bool reader_active = reader_end > 0 && std::chrono::steady_clock::now() - reader_seen < std::chrono::milliseconds(20);
        if (reader_active && len < reader_end + 8 * reader_block) return true;
        if (in_flight() < 2) extend_to(std::min(len + 2048, t));
        return len < t;
    }
    // the store should soon be longer than its buffers are (callers hold mu)
    bool wants_growth(size_t end) const { return cap < cap_max && !grow_requested && std::max(end, target) + (size_t)(30.0 * os_sr) > cap; }
    // the recorded fallback events ([0] = count, [1 + k] = sample index), one transfer however many engines ask
    // No wait for the store's stream (the background extension keeps it busy for seconds at a time): whoever asks stands at a t whose
    // samples a FINISHED launch produced, so the events below t are in memory; entries a running launch has counted but not yet written
    // read as ~0 (the list is initialised to all ones) and lie above every t.
    std::vector<unsigned long long> be_events() {
        std::vector<unsigned long long> h(1 + OW_TRAJ_BE_CAP, 0ull);
        if (hipMemcpy(h.data(), d_be, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost) != hipSuccess) std::fill(h.begin(), h.end(), 0ull);
        return h;
    }
    // fallback count of an engine standing at sample t (diag): the settle's own + the recorded events below t
    uint64_t be_count_at(long long t, const std::vector<unsigned long long>& h) const {
        uint64_t n = be_settle;
        const size_t k = (size_t)std::min<unsigned long long>(h[0], OW_TRAJ_BE_CAP);
        for (size_t i = 0; i < k; ++i) n += h[1 + i] < (unsigned long long)std::max<long long>(t, 0);
        if (h[0] > OW_TRAJ_BE_CAP) n += h[0] - OW_TRAJ_BE_CAP;     // beyond the list: counted, not placed
        return n;
    }
    uint64_t be_count_at(long long t) { return be_count_at(t, be_events()); }
};
std::mutex g_traj_mu;
// The registry is LEAKED on purpose (never destroyed): a static map's destructor would run ~TremTraj -- HIP calls -- during exit(),
// after the HIP runtime and any profiler attached to it have begun to shut down (rocprofv3 aborts there and the process hangs).  Stores
// die when ow_test_clear_settle_caches drops them and the last pool lets go; at process exit the driver reclaims the memory.
using TrajMap = std::map<std::pair<int, uint64_t>, std::shared_ptr<TremTraj>>;
TrajMap& traj_registry() { static TrajMap* m = new TrajMap(); return *m; }

// ow_tremolo_configure: capacity / lead of the stores of a device (seconds of audio at the chain rate); what is not configured comes from
// OW_TREM_TRAJ_SECONDS / OW_TREM_TRAJ_LEAD_SECONDS, then the defaults.
struct TrajConfig { double seconds = 0.0, lead = -1.0; };
std::map<int, TrajConfig> g_traj_cfg;      // guarded by g_traj_mu
constexpr double OW_TRAJ_DEFAULT_SECONDS = 1800.0, OW_TRAJ_DEFAULT_LEAD = 60.0, OW_TRAJ_FIRST_SECONDS = 150.0;

// Buffers grow on a helper thread (hipMalloc is no work for a thread that renders): cover() asks when the store comes within lead + 30 s
// of the end of its buffers, the helper doubles them.  One thread per process, started with the first store, parked on a condition
// variable; leaked like the registry (it must not touch the runtime while the process exits).
struct TrajGrower {
    // fixed tables: feed() / ask() are called from render, and the thread's own rounds run while hosts audit allocations -- nothing here
    // touches the heap (weak_ptr copies only count references)
    static constexpr int CAP = 16;
    std::mutex mu;
    std::condition_variable cv;
    std::weak_ptr<TremTraj> todo[CAP];     // stores whose buffers should double
    std::weak_ptr<TremTraj> fed[CAP];      // stores on their way to their target
    int n_todo = 0, n_fed = 0;
    bool stopping = false, parked = false;
    static bool same(const std::weak_ptr<TremTraj>& a, const std::weak_ptr<TremTraj>& b) { return !a.owner_before(b) && !b.owner_before(a); }
    void ask(const std::shared_ptr<TremTraj>& t) {
        { std::lock_guard<std::mutex> lk(mu); if (n_todo < CAP) todo[n_todo++] = t; }
        cv.notify_one();
    }
    void feed(const std::shared_ptr<TremTraj>& t) {
        {
            std::lock_guard<std::mutex> lk(mu);
            const std::weak_ptr<TremTraj> w(t);
            for (int i = 0; i < n_fed; ++i) if (same(fed[i], w)) return;          // already on the list
            if (n_fed == CAP) return;                                            // (sixteen chain rates x devices being fed at once: the renders' own block ahead remains)
            fed[n_fed++] = w;
        }
        cv.notify_one();
    }
    void stop() {       // from the exit handler: the thread makes no further runtime call once this returns (or after 2 s)
        std::unique_lock<std::mutex> lk(mu);
        stopping = true;
        cv.notify_all();
        cv.wait_for(lk, std::chrono::seconds(2), [&] { return parked; });
    }
    void run() {
        for (;;) {
            std::weak_ptr<TremTraj> grow[CAP], feeding[CAP];
            int n_grow = 0, n_feeding = 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (stopping) { parked = true; cv.notify_all(); for (;;) cv.wait(lk); }
                if (n_fed == 0) cv.wait(lk, [&] { return stopping || n_todo > 0 || n_fed > 0; });
                else cv.wait_for(lk, std::chrono::milliseconds(1));      // a launch of 2 048 steps lasts ~5 ms: look again soon
                if (stopping) { parked = true; cv.notify_all(); for (;;) cv.wait(lk); }
                for (int i = 0; i < n_todo; ++i) { grow[n_grow++] = todo[i]; todo[i].reset(); }
                n_todo = 0;
                for (int i = 0; i < n_fed; ++i) feeding[n_feeding++] = fed[i];
            }
            for (int i = 0; i < n_grow; ++i)
                if (std::shared_ptr<TremTraj> t = grow[i].lock()) {
                    try { size_t c; { std::lock_guard<std::mutex> lk(t->mu); c = t->cap; } t->grow_to(c * 2); }
                    catch (const std::exception& ex) { (void)hipGetLastError(); std::lock_guard<std::mutex> lk(t->mu); t->cap_max = t->cap; t->grow_requested = false;
                                                       std::fprintf(stderr, "openwurli-hip: tremolo trajectory store stays at %zu samples (%s)\n", t->cap, ex.what()); }
                }
            bool more[CAP] = {false};
            for (int i = 0; i < n_feeding; ++i)
                if (std::shared_ptr<TremTraj> t = feeding[i].lock()) {
                    try {
                        if (hipSetDevice(t->device) == hipSuccess) more[i] = t->feed_step();
                        bool ask_grow = false;
                        { std::lock_guard<std::mutex> lk(t->mu); if (t->wants_growth(t->len)) { t->grow_requested = true; ask_grow = true; } }
                        if (ask_grow) { std::lock_guard<std::mutex> lk(mu); if (n_todo < CAP) todo[n_todo++] = t; }
                    } catch (const std::exception&) { (void)hipGetLastError(); more[i] = false; }
                }
            {
                std::lock_guard<std::mutex> lk(mu);
                // stores that were added while this round ran stay; the ones this round finished go
                int k = 0;
                for (int i = 0; i < n_fed; ++i) {
                    bool was = false, stays = false;
                    for (int j = 0; j < n_feeding; ++j) if (same(fed[i], feeding[j])) { was = true; stays = more[j]; }
                    if (!was || stays) { if (k != i) fed[k] = fed[i]; ++k; }
                }
                for (int i = k; i < n_fed; ++i) fed[i].reset();
                n_fed = k;
            }
        }
    }
};
TrajGrower& traj_grower() {
    static TrajGrower* g = [] {
        TrajGrower* x = new TrajGrower();
        std::thread([x] { x->run(); }).detach();
        // exit(): handlers run in reverse order of registration, so this one -- registered long after the HIP runtime's own -- parks the
        // thread before the runtime starts to come down (a launch or an event query during teardown crashes the process)
        std::atexit([] { traj_grower().stop(); });
        return x;
    }();
    return *g;
}

// the store of (device, hc.os_sr), created (and settled) on first use.  n_engines: size of the pool that asks -- a big pool reserves the
// whole configured capacity at once (render never waits for a growth), a small one starts with 150 s (115 MB at 96 kHz) and grows.
std::shared_ptr<TremTraj> traj_acquire(int device, const OwConsts& hc, const OwConsts& k48, bool use_settle_cache, size_t n_engines = 1) {
    uint64_t rate_bits; std::memcpy(&rate_bits, &hc.os_sr, 8);
    const std::pair<int, uint64_t> key(device, rate_bits);
    std::shared_ptr<TremTraj> t;
    {
    std::lock_guard<std::mutex> lk(g_traj_mu);
    TrajMap& g_traj = traj_registry();
    auto it = g_traj.find(key);
    if (it != g_traj.end()) t = it->second;
    else {
    t = std::make_shared<TremTraj>();
    t->device = device; t->os_sr = hc.os_sr;
    TrajConfig cfg;
    { auto c = g_traj_cfg.find(device); if (c != g_traj_cfg.end()) cfg = c->second; }
    double seconds = cfg.seconds > 0.0 ? cfg.seconds : OW_TRAJ_DEFAULT_SECONDS, lead = cfg.lead >= 0.0 ? cfg.lead : OW_TRAJ_DEFAULT_LEAD;
    if (!(cfg.seconds > 0.0)) if (const char* env = std::getenv("OW_TREM_TRAJ_SECONDS")) { const double v = std::atof(env); if (v > 0.0) seconds = v; }
    if (!(cfg.lead >= 0.0)) if (const char* env = std::getenv("OW_TREM_TRAJ_LEAD_SECONDS")) { const double v = std::atof(env); if (v >= 0.0) lead = v; }
    const double want = std::min(seconds * hc.os_sr, 4.0e9);
    t->cap_max = ((size_t)want + OW_TRAJ_CK - 1) / OW_TRAJ_CK * OW_TRAJ_CK;
    t->lead = (size_t)std::min(lead * hc.os_sr, (double)t->cap_max);
    const size_t first = n_engines >= 4096 ? t->cap_max : std::min(t->cap_max, ((size_t)(OW_TRAJ_FIRST_SECONDS * hc.os_sr) + OW_TRAJ_CK - 1) / OW_TRAJ_CK * OW_TRAJ_CK);
    t->cap = first;
    HIP_OK(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    for (auto& m : t->mark) HIP_OK(hipEventCreateWithFlags(&m.ev, hipEventDisableTiming));
    HIP_OK(hipMalloc(&t->dK, sizeof(OwConsts)));
    HIP_OK(hipMalloc(&t->d_r, sizeof(double) * (t->cap + 64)));
    HIP_OK(hipMalloc(&t->d_state, sizeof(double) * 18));
    HIP_OK(hipMalloc(&t->d_ckpt, sizeof(double) * TremTraj::ckpt_doubles(t->cap)));
    HIP_OK(hipMalloc(&t->d_be, sizeof(unsigned long long) * (1 + OW_TRAJ_BE_CAP)));
    HIP_OK(hipMalloc(&t->d_zero, sizeof(uint32_t)));
    HIP_OK(hipMemsetAsync(t->d_be, 0xFF, sizeof(unsigned long long) * (1 + OW_TRAJ_BE_CAP), t->stream));
    HIP_OK(hipMemsetAsync(t->d_be, 0, sizeof(unsigned long long), t->stream));
    HIP_OK(hipMemsetAsync(t->d_zero, 0, sizeof(uint32_t), t->stream));
    HIP_OK(hipMemcpyAsync(t->dK, &hc, sizeof(OwConsts), hipMemcpyHostToDevice, t->stream));
    DevMem dk48;
    dk48.alloc(sizeof(OwConsts));
    HIP_OK(hipMemcpyAsync(dk48.p, &k48, sizeof(OwConsts), hipMemcpyHostToDevice, t->stream));
    const TremSettled ts = trem_settled_rows(device, hc.os_sr, t->dK, dk48.as<OwConsts>(), t->d_state, t->d_zero, t->stream, use_settle_cache);
    std::memcpy(&t->be_settle, &ts.rows[17], 8);
    HIP_OK(hipMemsetAsync(t->d_state + 17, 0, sizeof(double), t->stream));   // the trajectory's own events go to d_be
    HIP_OK(hipStreamSynchronize(t->stream));
    g_traj[key] = t;
    traj_grower();                                                        // the helper thread exists before any render could need it
    }
    }
    // (outside the registry lock) a big pool on a store that was created small: all of it now, where allocating is allowed
    if (n_engines >= 4096) t->grow_to(t->cap_max);
    // a small pool can outrun the one oscillator: it starts to run ahead right away, fed by the helper thread, so that a host that never
    // calls ow_tremolo_prefetch finds its first blocks' samples waiting.  (A big pool renders far slower than the oscillator steps and
    // only ever needs the block ahead that its own renders enqueue.)
    if (n_engines < 4096) {
        bool feed = false;
        { std::lock_guard<std::mutex> lk(t->mu); t->cover(0, 0, t->lead, &feed); }
        if (feed) traj_grower().feed(t);
    }
    return t;
}

enum { INIT_NEW = 1, INIT_RATE = 2, INIT_RESET = 0 };

// Job paths (batch render, render-midi): a quad of lanes per preamp state (ow_chain_wide.h) while the jobs are too few to fill the
// chip with one lane pair each -- their run time is then the chain's serial latency.  OW_CHAIN_WIDE=0/1 forces the choice (the
// parity test compares the two kernels bit for bit).
static inline bool chain_wide(size_t n_jobs) {
    if (const char* env = std::getenv("OW_CHAIN_WIDE")) return env[0] == '1';
    return n_jobs <= 8192;
}
// OW_JOB_FUSED=0: the quad job chain as one wavefront (k_job_chain_wide) instead of preamp | output stage on two (k_job_chain_fused)
// (two wavefronts of 400+ registers per eight jobs: 512 workgroups fill the chip, more of them take a second round -- 8 192 jobs measured
// 319 against 191 ms -- so the fused form is for up to 4 096 jobs)
static inline bool job_chain_fused(size_t n_jobs) {
    if (const char* env = std::getenv("OW_JOB_FUSED")) return env[0] != '0';
    return n_jobs <= 4096;
}
// The batch render may run the voices of its jobs BESIDE this chain (ow_batch_render: k_job_voice on a second stream publishes its
// progress, k_job_chain_fused waits chunk by chunk): only while the chain's workgroups leave SIMDs free for the voice kernel -- 2 048 jobs
// are 256 workgroups of two one-per-SIMD wavefronts, half the chip -- so that the producer can always be scheduled.
static inline bool job_voice_overlap(size_t n_jobs) {
    if (const char* env = std::getenv("OW_JOB_OVERLAP")) { if (env[0] == '0') return false; }
    return chain_wide(n_jobs) && job_chain_fused(n_jobs) && n_jobs <= 2048;
}
// OW_JOB_ROW=0: the fused job chain with a quad per solver state (k_job_chain_fused) instead of a row of sixteen lanes (k_job_chain_row:
// five wavefronts per eight jobs -- while they all find a SIMD of their own beside the voice kernel)
static inline bool job_chain_row(size_t n_jobs) {
    if (const char* env = std::getenv("OW_JOB_ROW")) return env[0] != '0';
    return n_jobs <= 1024;
}
static void launch_job_chain_legacy(const OwConsts* dK, const owdev::OwJobDev* d_jobs, const double* d_in, double* d_out, size_t n_jobs, long long n,
                                    long long stride, hipStream_t st, const int* voice_prog = nullptr) {
    if (voice_prog && !(chain_wide(n_jobs) && job_chain_fused(n_jobs))) throw std::runtime_error("job chain: overlap with the voices needs the fused chain");
    if (chain_wide(n_jobs) && job_chain_fused(n_jobs) && job_chain_row(n_jobs))
        owdev::k_job_chain_row<<<dim3((unsigned)((n_jobs + 7) / 8)), dim3(320), 0, st>>>(dK, d_jobs, d_in, d_out, (int)n_jobs, n, stride, voice_prog,
                                                                                         voice_prog ? const_cast<int*>(voice_prog) + (n_jobs + 63) / 64 : nullptr);
    else if (chain_wide(n_jobs) && job_chain_fused(n_jobs))
        owdev::k_job_chain_fused<<<dim3((unsigned)((n_jobs + 7) / 8)), dim3(128), 0, st>>>(dK, d_jobs, d_in, d_out, (int)n_jobs, n, stride, voice_prog,
                                                                                           voice_prog ? const_cast<int*>(voice_prog) + (n_jobs + 63) / 64 : nullptr);
    else if (chain_wide(n_jobs))
        owdev::k_job_chain_wide<<<dim3((unsigned)((n_jobs + 7) / 8)), dim3(64), 0, st>>>(dK, d_jobs, d_in, d_out, (int)n_jobs, n, stride);
    else
        owdev::k_job_chain<false><<<dim3((unsigned)((n_jobs + 31) / 32)), dim3(64), 0, st>>>(dK, d_jobs, d_in, d_out, nullptr, (int)n_jobs, n, stride);
}

ow_pool* pool_create(double sample_rate, size_t n_engines, int device, int preamp_kind, int power_amp_kind, int tremolo_kind, bool voices_only, bool no_traj = false);
void pool_destroy(ow_pool* p);
void pa_settled_to_device(int device, double* d_dst, hipStream_t st);

// The chain of the job paths (`preamp-bench render` / `render-midi`, tools/preamp-bench/src/main.rs:413-497, 1880-1890) for n_jobs rows
// of voice signal d_in -> d_out (both [n_jobs][stride]), with everything the commands' flags can ask for:
//   * --tremolo-depth > 0 on some job: ONE Twin-T stream for the call (Tremolo::new settles every job's oscillator to the same state,
//     and the oscillator takes no input), produced by the product's own tremolo kernel from a freshly built pool of one; each job
//     applies its own depth divider;
//   * the melange preamp (`--features melange-preamp` build) through k_job_chain<true>;
//   * the melange power amp (a build without `legacy-power-amp`): PowerAmp::new() is new_at_sample_rate(44 100) whatever the render's
//     rate is (power_amp.rs:321-323), it runs at the BASE rate on preamp x volume^2 -- as its own launch (eight lanes per job,
//     k_mpa_debug) between the chain kernel and the speaker stage.
struct JobChainCfg { double sample_rate; int device, preamp_kind, power_amp_kind, no_rail_sag; };
// true when run_job_chain will take the plain legacy chain (launch_job_chain_legacy) for these jobs
static bool job_chain_is_plain_legacy(const JobChainCfg& cfg, const std::vector<owdev::OwJobDev>& hj) {
    if (cfg.preamp_kind != OW_PREAMP_LEGACY8) return false;
    bool any_pa = false;
    for (const auto& j : hj) {
        if ((j.tremolo_depth > 0.0 && !j.no_preamp) || j.no_preamp) return false;
        any_pa = any_pa || j.poweramp;
    }
    return !(cfg.power_amp_kind == OW_POWER_AMP_MELANGE && any_pa);
}
void run_job_chain(const JobChainCfg& cfg, const OwConsts* dK, const std::vector<owdev::OwJobDev>& hj, const owdev::OwJobDev* d_jobs, const double* d_in,
                   double* d_out, size_t n_jobs, long long n, long long stride, hipStream_t st, const int* voice_prog = nullptr) {
    if (cfg.preamp_kind != OW_PREAMP_LEGACY8 && cfg.preamp_kind != OW_PREAMP_MELANGE12) throw std::runtime_error("unknown preamp_kind");
    if (cfg.power_amp_kind != OW_POWER_AMP_BEHAVIORAL && cfg.power_amp_kind != OW_POWER_AMP_MELANGE) throw std::runtime_error("unknown power_amp_kind");
    bool any_trem = false, any_special = false, any_pa = false;
    for (const auto& j : hj) {
        any_trem = any_trem || (j.tremolo_depth > 0.0 && !j.no_preamp);
        any_special = any_special || j.no_preamp;
        any_pa = any_pa || j.poweramp;
    }
    const bool mpa = cfg.power_amp_kind == OW_POWER_AMP_MELANGE && any_pa;
    DevMem d_r, d_settled, d_att, d_amp, d_pac, d_pas;
    std::shared_ptr<TremTraj> traj;
    const double* trem = nullptr;
    if (any_trem) {
        // Tremolo::new(depth, preamp rate) without a warm-up: every job's cell starts at t = 0 of the shared trajectory of this chain rate
        const bool os = cfg.sample_rate < 88200.0;
        const long long n_os = n * (os ? 2 : 1);
        const Switches sw = Switches::from_env();          // offline entry point: read once per call
        if (sw.trem_traj) {
            std::unique_ptr<OwConsts> hc(new OwConsts()), k48(new OwConsts());
            owhip::build_consts(*hc, cfg.sample_rate, OW_PREAMP_LEGACY8);
            owhip::build_consts(*k48, 24000.0, OW_PREAMP_LEGACY8);
            traj = traj_acquire(cfg.device, *hc, *k48, sw.trem_cache);
            if ((size_t)n_os <= traj->cap_max) {
                traj->grow_to((size_t)n_os);                 // (offline entry point: allocating here is fine; no-op when the buffers reach that far)
                hipEvent_t ev;
                { std::lock_guard<std::mutex> lk(traj->mu); ev = traj->cover((size_t)n_os, 0); trem = traj->d_r; }
                if (ev) HIP_OK(hipStreamWaitEvent(st, ev, 0));
            }
        }
        if (!trem) {                                        // longer than the store: one oscillator for this call, from a pool of one
            d_r.alloc(sizeof(double) * (size_t)n_os);
            struct PoolGuard { ow_pool* p; ~PoolGuard() { pool_destroy(p); } } g{pool_create(cfg.sample_rate, 1, cfg.device, OW_PREAMP_LEGACY8, OW_POWER_AMP_BEHAVIORAL,
                                                                                                 OW_TREMOLO_TWIN_T, false, /*no_traj=*/true)};
            // the fresh pool's oscillator rows are Tremolo::new's settled state; n_os steps of Tremolo::process, R written per step
            owdev::k_tremolo_wide<false><<<dim3(1), dim3(64), 0, g.p->stream>>>(g.p->dK, g.p->d_cs, d_r.as<double>(), 1, n_os, g.p->d_leaders, 1);
            HIP_OK(hipGetLastError());
            HIP_OK(hipStreamSynchronize(g.p->stream));
            trem = d_r.as<double>();
        }
    }
    double* chain_out = d_out;
    if (mpa) {
        d_att.alloc(sizeof(double) * n_jobs * (size_t)stride);
        d_amp.alloc(sizeof(double) * n_jobs * (size_t)stride);
        chain_out = d_att.as<double>();
    }
    const int out_mode = mpa ? owdev::JOB_OUT_PA_INPUT : owdev::JOB_OUT_FINAL;
    if (cfg.preamp_kind == OW_PREAMP_MELANGE12) {
        d_settled.alloc(sizeof(double) * 18);
        mel_settled_to_device(cfg.device, d_settled.as<double>(), st);
        owdev::k_job_chain<true><<<dim3((unsigned)((n_jobs + 31) / 32)), dim3(64), 0, st>>>(dK, d_jobs, d_in, chain_out, d_settled.as<double>(), (int)n_jobs, n, stride,
                                                                                            trem, out_mode);
    } else if (!any_trem && !any_special && !mpa) {
        launch_job_chain_legacy(dK, d_jobs, d_in, chain_out, n_jobs, n, stride, st, voice_prog);
    } else {
        owdev::k_job_chain<false><<<dim3((unsigned)((n_jobs + 31) / 32)), dim3(64), 0, st>>>(dK, d_jobs, d_in, chain_out, nullptr, (int)n_jobs, n, stride, trem, out_mode);
    }
    HIP_OK(hipGetLastError());
    if (mpa) {
        std::unique_ptr<OwPaConsts> hpa(new OwPaConsts());
        owhip::build_pa_consts(*hpa, 44100.0);
        d_pac.alloc(sizeof(OwPaConsts));
        d_pas.alloc(sizeof(double) * owdev::PAS_CIRCUIT_END);
        pa_settled_to_device(cfg.device, d_pas.as<double>(), st);
        HIP_OK(hipMemcpyAsync(d_pac.p, hpa.get(), sizeof(OwPaConsts), hipMemcpyHostToDevice, st));
        owdev::k_mpa_debug<<<dim3((unsigned)((n_jobs + PA_EPB - 1) / PA_EPB)), dim3(PA_WPB * 64), 0, st>>>(d_pac.as<OwPaConsts>(), d_pas.as<double>(), d_att.as<double>(),
                                                                                                       d_amp.as<double>(), nullptr, n, (int)n_jobs, cfg.no_rail_sag ? 0 : 1,
                                                                                                       nullptr, nullptr, nullptr, stride);
        owdev::k_job_speaker<<<dim3((unsigned)((n_jobs + 63) / 64)), dim3(64), 0, st>>>(dK, d_jobs, d_att.as<double>(), d_amp.as<double>(), d_out, (int)n_jobs, n, stride);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(st));      // hpa and the staging buffers go out of scope
    }
    HIP_OK(hipStreamSynchronize(st));
}

// chain (re)initialisation of engines [e0, e0+ne): DC states on the device, then the Twin-T settle
// (50 warm-up steps at the codegen matrices + 2 s at the pool rate), all in the product kernels.
// Pools this small leave SIMDs idle, and the oscillator's serial latency is their block time: four lanes per engine (ow_trem_wide.h).
// Melange power amp: engines dispatched by falling demand (OW_PA_SORT=0: in index order -- the same samples, tested)
static bool power_amp_ordered(const ow_pool* p) { return p->sw.pa_sort != 0; }
// engines of k_post_mpa the chip holds at once: two workgroups of PA_EPB per CU (LDS)
static int power_amp_resident_engines(const ow_pool* p) {
    if (p->sw.pa_sort == 2) return PA_EPB;      // '2': order every block of more than one workgroup (tests)
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || cus <= 0) cus = 256;
    return cus * 2 * PA_EPB;
}
// OW_TREM_WIDE=0/1 forces the choice (the parity test compares the two kernels bit for bit).
static inline bool trem_wide(const ow_pool* p, int ne) { return p->sw.trem_wide >= 0 ? p->sw.trem_wide == 1 : ne <= 16384; }
// lane = group tremolo kernel at pool scale, measured at 131 072 engines / oscillators: serialised in front of the voices 31.8 ms per
// block (voices 12.8, tremolo 9.1 on an empty chip), overlapped 29.7 ms -- each kernel alone leaves issue slots the other can use.  So
// the default stays overlapped; the serialised schedule is how the two kernels' own times are measured (profiles/).
static inline bool trem_serialised(const ow_pool* p) { return p->sw.trem_serial; }
// legacy preamp with a quad per solver state (k_preamp_wide): while the pool leaves most SIMDs empty the kernel's time is the serial
// latency of one sample, which the quad shortens; beyond ~4 096 engines the lane-pair kernel's lower instruction count wins
static inline bool preamp_wide(const ow_pool* p, int ne) { return p->sw.preamp_wide >= 0 ? p->sw.preamp_wide == 1 : ne <= 4096; }
// ... and the output stage in the same launch behind it (k_chain_fused): legacy preamp + behavioural power amp only
static inline bool chain_fused(const ow_pool* p, int ne) {
    if (p->hc.preamp_kind != OW_PREAMP_LEGACY8 || p->power_amp_kind != OW_POWER_AMP_BEHAVIORAL) return false;
    return p->sw.chain_fused >= 0 ? p->sw.chain_fused == 1 : preamp_wide(p, ne);
}
// ... with the row step (k_chain_row): while every preamp wavefront (two engines) has a SIMD of its own
static inline bool chain_row(const ow_pool* p, int ne) { return p->sw.chain_row >= 0 ? p->sw.chain_row == 1 : ne <= 1024; }
// ---- tremolo phase groups (see ow_pool) ---------------------------------------------------------------------------------
void trem_groups_changed(ow_pool* p) {
    HIP_OK(hipMemcpyAsync(p->d_lead, p->h_lead, sizeof(uint32_t) * p->I, hipMemcpyHostToDevice, p->stream));
    p->lead_list_valid = false;
    p->split_e0 = p->split_ne = -1;
}
// After this no group has members on both sides of the boundary of [e0, e0+ne): a straddling group is cut in two, the part that
// loses the leader gets its lowest member as the new leader together with a copy of the 18 tremolo rows.  Init / reset / warm-up of
// a sub-range then only touch groups that lie wholly inside it.  O(I) host work, on reset-class calls only.
void trem_split_at_range(ow_pool* p, int e0, int ne) {
    const uint32_t I = (uint32_t)p->I, lo = (uint32_t)e0, hi = (uint32_t)(e0 + ne);
    if (ne >= (int)I || (p->split_e0 == e0 && p->split_ne == ne)) return;
    const uint32_t NONE = 0xFFFFFFFFu;
    bool straddle = false;
    std::fill(p->grp_in.begin(), p->grp_in.end(), NONE);
    std::fill(p->grp_out.begin(), p->grp_out.end(), NONE);
    for (uint32_t e = 0; e < I; ++e) {
        const uint32_t x = p->h_lead[e];
        uint32_t& slot = (e >= lo && e < hi) ? p->grp_in[x] : p->grp_out[x];
        if (slot == NONE) slot = e;                    // ascending scan: the first one found is the lowest
    }
    size_t n_copy = 0;
    for (uint32_t x = 0; x < I; ++x) {
        if (p->grp_in[x] == NONE || p->grp_out[x] == NONE) continue;
        straddle = true;
        const bool x_inside = x >= lo && x < hi;
        const uint32_t fresh = x_inside ? p->grp_out[x] : p->grp_in[x];     // leader of the part that loses x
        p->h_copy[n_copy] = x; p->h_copy[I + n_copy] = fresh; ++n_copy;
    }
    if (straddle) {
        invalidate_spec(p);                             // the oscillator rows must be the committed ones before they are copied
        HIP_OK(hipStreamSynchronize(p->stream));
        for (uint32_t e = 0; e < I; ++e) {
            const uint32_t x = p->h_lead[e];
            if (p->grp_in[x] == NONE || p->grp_out[x] == NONE) continue;
            const bool e_inside = e >= lo && e < hi, x_inside = x >= lo && x < hi;
            if (e_inside != x_inside) p->h_lead[e] = x_inside ? p->grp_out[x] : p->grp_in[x];
        }
        HIP_OK(hipMemcpyAsync(p->d_copy, p->h_copy, sizeof(uint32_t) * n_copy, hipMemcpyHostToDevice, p->stream));
        HIP_OK(hipMemcpyAsync(p->d_copy + I, p->h_copy + I, sizeof(uint32_t) * n_copy, hipMemcpyHostToDevice, p->stream));
        owdev::k_trem_copy_rows<<<dim3((unsigned)((n_copy + 63) / 64)), dim3(64), 0, p->stream>>>(p->d_cs, (int)I, p->d_copy, p->d_copy + I, (int)n_copy);
        trem_groups_changed(p);
        HIP_OK(hipStreamSynchronize(p->stream));
    }
    p->split_e0 = e0; p->split_ne = ne;
}
// Compact list of the group leaders inside [e0, e0+ne) (groups do not straddle the range: trem_split_at_range) -> d_leaders.
// Engines on the shared trajectory have no oscillator of their own and are not listed.
void trem_leader_list(ow_pool* p, int e0, int ne) {
    if (p->lead_list_valid && p->lead_e0 == e0 && p->lead_ne == ne) return;
    if (p->stream_trem) HIP_OK(hipStreamSynchronize(p->stream_trem));   // a kernel still reading the old list (rare path: groups or range changed)
    int n = 0;
    const bool traj = p->traj != nullptr;
    for (int k = 0; k < ne; ++k)
        if (p->h_lead[e0 + k] == (uint32_t)(e0 + k) && !(traj && p->h_birth[e0 + k] != OW_OFF_TRAJ)) p->h_leaders[n++] = (uint32_t)(e0 + k);
    HIP_OK(hipMemcpy(p->d_leaders, p->h_leaders, sizeof(uint32_t) * std::max(n, 1), hipMemcpyHostToDevice));
    p->n_lead = n; p->lead_e0 = e0; p->lead_ne = ne; p->lead_list_valid = true;
}

// min over the engines on the trajectory of birth[e] (the oldest decides how far the store must reach); O(I), reset-class calls only
void traj_recount(ow_pool* p) {
    long long mn = p->trem_clock; size_t n = 0;
    for (size_t e = 0; e < p->I; ++e) if (p->h_birth[e] != OW_OFF_TRAJ) { mn = std::min(mn, p->h_birth[e]); ++n; }
    p->min_birth = mn; p->n_on_traj = n;
}
void traj_upload_births(ow_pool* p, int e0, int ne) {     // synchronous: h_birth is pageable and changes again right away
    HIP_OK(hipStreamSynchronize(p->stream));              // a k_trem_birth_shift of a sub-range render still queued would shift the new values again
    HIP_OK(hipMemcpy(p->d_birth + e0, p->h_birth.data() + e0, sizeof(long long) * (size_t)ne, hipMemcpyHostToDevice));
}

// Engines of [e0, e0+ne) whose next block would run past the end of the store leave the trajectory: each gets its own oscillator rows,
// rebuilt from the checkpoint below its t and stepped up to it (k_trem_from_ckpt, < OW_TRAJ_CK steps), and is a phase group of one from
// here on (per-group oscillator, as in rounds 1-3).  Cold: once in an engine's life, after OW_TREM_TRAJ_SECONDS without a reset.
void trem_evict(ow_pool* p, int e0, int ne, int n_os) {
    TremTraj* T = p->traj.get();
    std::vector<uint32_t> eng; std::vector<long long> tp;
    long long tmax = 0;
    for (int k = 0; k < ne; ++k) {
        const long long b = p->h_birth[e0 + k];
        if (b == OW_OFF_TRAJ || (size_t)(p->trem_clock - b) + (size_t)n_os <= T->cap_max) continue;
        eng.push_back((uint32_t)(e0 + k)); tp.push_back(p->trem_clock - b); tmax = std::max(tmax, p->trem_clock - b);
    }
    if (eng.empty()) return;
    invalidate_spec(p);
    HIP_OK(hipStreamSynchronize(p->stream));
    hipEvent_t ev;
    { std::lock_guard<std::mutex> lk(T->mu); ev = T->cover((size_t)tmax, 0); }
    if (ev) HIP_OK(hipEventSynchronize(ev));
    std::vector<unsigned long long> be(eng.size());
    {
        const std::vector<unsigned long long> events = T->be_events();
        for (size_t i = 0; i < eng.size(); ++i) be[i] = T->be_count_at(tp[i], events);
    }
    // scratch reserved at pool creation (d_evict: [I] engines as u64 | [I] t | [I] fallback counts): no allocation on this path
    uint32_t* de = (uint32_t*)p->d_evict; long long* dt = (long long*)(p->d_evict + p->I); unsigned long long* db = p->d_evict + 2 * p->I;
    HIP_OK(hipMemcpy(de, eng.data(), sizeof(uint32_t) * eng.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dt, tp.data(), sizeof(long long) * eng.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db, be.data(), sizeof(unsigned long long) * eng.size(), hipMemcpyHostToDevice));
    const double* tr; const double* tc;
    { std::lock_guard<std::mutex> lk(T->mu); tr = T->d_r; tc = T->d_ckpt; }
    owdev::k_trem_from_ckpt<<<dim3((unsigned)((eng.size() + 63) / 64)), dim3(64), 0, p->stream>>>(p->dK, tr, tc, p->d_cs, (int)p->I, de, dt, db, (int)eng.size());
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(p->stream));
    for (uint32_t e : eng) { p->h_birth[e] = OW_OFF_TRAJ; p->h_lead[e] = e; }
    traj_upload_births(p, e0, ne);
    traj_recount(p);
    trem_groups_changed(p);
    HIP_OK(hipStreamSynchronize(p->stream));
}

void chain_init_range(ow_pool* p, int e0, int ne, int mode, const std::vector<double>& depth0) {
    invalidate_spec(p);
    HIP_OK(hipStreamSynchronize(p->stream));
    const int I = (int)p->I;
    const bool on_traj = p->traj != nullptr;
    if (on_traj) {
        // Tremolo::new / reset: the cell starts at t = 0 of the shared trajectory (its settled state IS Tremolo::new's) -- nothing to settle,
        // nothing to copy; an engine that had left the trajectory for its own oscillator comes back
        for (int k = 0; k < ne; ++k) {
            if (p->h_birth[e0 + k] == OW_OFF_TRAJ) p->lead_list_valid = false;
            p->h_birth[e0 + k] = p->trem_clock;
            p->h_lead[e0 + k] = (uint32_t)(e0 + k);
        }
        traj_upload_births(p, e0, ne);
        traj_recount(p);
    } else {
        // the range's engines get identical fresh tremolo states below: cut them out of the groups they shared with engines outside the
        // range (those keep their oscillator), then make the range one group led by its first engine
        trem_split_at_range(p, e0, ne);
        for (int k = 0; k < ne; ++k) p->h_lead[e0 + k] = (uint32_t)e0;
        trem_groups_changed(p);
        p->split_e0 = e0; p->split_ne = ne;
        trem_leader_list(p, e0, ne);
    }
    if (mode == INIT_RESET) {
        // reset() snaps every smoother to ITS target (engine.rs:245-249), and LinearSmoother::set_target stores the target at once
        // (engine.rs:86-99) -- also for a setter call the device has not seen yet because no block was rendered since.  Hand the
        // host targets to the kernel; the retarget requests themselves are dropped by engine_host_reset.
        for (int k = 0; k < ne; ++k) {
            const ow_engine* en = p->engines[e0 + k];
            p->h_snap[0 * p->I + e0 + k] = en->depth.target; p->h_snap[1 * p->I + e0 + k] = en->spk.target; p->h_snap[2 * p->I + e0 + k] = en->volume.target;
        }
        for (int r = 0; r < 3; ++r)
            HIP_OK(hipMemcpyAsync(p->d_snap + (size_t)r * p->I + e0, p->h_snap + (size_t)r * p->I + e0, sizeof(double) * ne, hipMemcpyHostToDevice, p->stream));
    }
    // depth0: one value per engine of the range (Tremolo::new(depth)); the kernel takes a scalar, so group equal values
    int i = 0;
    while (i < ne) {
        int j = i + 1;
        while (j < ne && depth0[j] == depth0[i]) ++j;
        owdev::k_chain_init<<<dim3((j - i + 63) / 64), dim3(64), 0, p->stream>>>(p->dK, p->d_cs, p->d_snap, I, e0 + i, j - i, mode, depth0[i]);
        i = j;
    }
    if (p->hc.preamp_kind == OW_PREAMP_MELANGE12)   // DkPreamp::new / reset of the melange adapter: settled state at the chain rate
        owdev::k_mel_init<<<dim3((2 * ne + 63) / 64), dim3(64), 0, p->stream>>>(p->d_cs, p->d_mel_settled, p->d_noise, I, e0, ne);
    if (p->power_amp_kind == OW_POWER_AMP_MELANGE)  // PowerAmp::new_at_sample_rate (new / set_sample_rate) or PowerAmp::reset (reset keeps last_good)
        owdev::k_mpa_init<<<dim3((ne + 63) / 64), dim3(64), 0, p->stream>>>(p->dPa, p->d_pa_settled, p->d_pa, I, e0, ne, mode != INIT_RESET ? 1 : 0);
    if (!on_traj && p->tremolo_kind != OW_TREMOLO_LEGACY_LFO) {   // (legacy-tremolo build: nothing to settle, the LFO starts at phase 0, k_chain_init)
        // the range is one phase group led by e0: Tremolo::new's settle (tremolo.rs:92-102) for that one oscillator
        if (p->sw.trem_cache) {
            trem_settled_rows(p->device, p->hc.os_sr, p->dK, p->dK48, p->d_trem_settled, p->d_zero, p->stream, true);
            owdev::k_trem_load_settled<<<dim3(1), dim3(64), 0, p->stream>>>(p->d_cs, I, e0, p->d_trem_settled);
        } else {
            const long long n_settle = (long long)owhip::sat_u32(p->hc.os_sr * 2.0);
            owdev::k_tremolo_wide<true><<<dim3(1), dim3(64), 0, p->stream>>>(p->dK48, p->d_cs, nullptr, I, 50LL, p->d_leaders, 1);
            owdev::k_tremolo_wide<true><<<dim3(1), dim3(64), 0, p->stream>>>(p->dK, p->d_cs, nullptr, I, n_settle, p->d_leaders, 1);
        }
    }
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(p->stream));   // the tremolo stream picks these rows up next (init-time sync)
}

void upload_consts(ow_pool* p, double sr, int preamp_kind) {
    invalidate_spec(p);
    owhip::build_consts(p->hc, sr, preamp_kind);
    p->hc.tremolo_kind = (uint32_t)p->tremolo_kind;
    OwConsts k48;
    owhip::build_consts(k48, 24000.0, preamp_kind);  // os_sr = 48 kHz -> codegen-rate tremolo matrices
    k48.tremolo_kind = (uint32_t)p->tremolo_kind;
    HIP_OK(hipMemcpyAsync(p->dK, &p->hc, sizeof(OwConsts), hipMemcpyHostToDevice, p->stream));
    HIP_OK(hipMemcpyAsync(p->dK48, &k48, sizeof(OwConsts), hipMemcpyHostToDevice, p->stream));
    // the shared trajectory of this (device, chain rate); the callers (pool_create, set_sample_rate) re-initialise every engine next
    p->traj.reset();
    if (!p->voices_only && p->tremolo_kind == OW_TREMOLO_TWIN_T && p->sw.trem_traj) {
        // a store that cannot be had (no room for it on a crowded device) is not an error: the pool runs one oscillator per phase
        // group, as under OW_TREM_TRAJ=0 -- the same samples
        try { p->traj = traj_acquire(p->device, p->hc, k48, p->sw.trem_cache, p->I); }
        catch (const std::exception& ex) { p->traj.reset(); (void)hipGetLastError(); std::fprintf(stderr, "openwurli-hip: no tremolo trajectory store (%s): per-group oscillators\n", ex.what()); }
    }
    if (!p->traj) { std::fill(p->h_birth.begin(), p->h_birth.end(), OW_OFF_TRAJ); if (p->d_birth) traj_upload_births(p, 0, (int)p->I); p->n_on_traj = 0; }
    if (p->power_amp_kind == OW_POWER_AMP_MELANGE) {
        std::unique_ptr<OwPaConsts> hpa(new OwPaConsts());
        owhip::build_pa_consts(*hpa, p->hc.os_sr);          // the amp runs at the chain rate (engine.rs:207-213)
        HIP_OK(hipMemcpyAsync(p->dPa, hpa.get(), sizeof(OwPaConsts), hipMemcpyHostToDevice, p->stream));
        HIP_OK(hipStreamSynchronize(p->stream));
    }
    HIP_OK(hipStreamSynchronize(p->stream));
}

// Host threads worth starting: the CPU count capped by the cgroup CPU quota (cpu.max "quota period").  A container limited
// to 16 CPUs on a 256-thread host gets throttled for whole scheduler periods when 64 threads burst at once.
static size_t effective_cpus() {
    static const size_t cached = [] {
    size_t n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long long period = 0;
        if (std::fscanf(f, "%31s %lld", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0) {
            const long long q = std::atoll(quota);
            if (q > 0) n = std::min<size_t>(n, (size_t)std::max<long long>(1, q / period));
        }
        std::fclose(f);
    }
    // one process per GPU (torchrun sets LOCAL_WORLD_SIZE): the ranks of a node share the quota, so each takes its share of the
    // threads -- eight ranks bursting 16 threads each into a 16-CPU quota get throttled for whole scheduler periods
    if (const char* lws = std::getenv("LOCAL_WORLD_SIZE")) { const long w = std::atol(lws); if (w > 1) n = std::max<size_t>(1, n / (size_t)w); }
    if (const char* env = std::getenv("OW_HOST_THREADS")) { const long v = std::atol(env); if (v >= 1 && v <= 256) n = (size_t)v; }
    return n;
    }();
    return cached;
}
size_t Workers::host_threads() { return std::min<size_t>(effective_cpus(), OW_MAX_SLICES); }

// see ow_pool::dev_ops_applied
void vm_settle_applied(ow_pool* p) {
    p->dev_ops_applied = false;
    const uint32_t lo = p->applied_lo, hi = p->applied_hi;
    const size_t T = hi - lo >= 16384 ? std::min<size_t>(effective_cpus(), 32) : 1;
    const uint32_t per = (uint32_t)((hi - lo + T - 1) / T);
    auto slice = [&](size_t t) {
        const uint32_t k1 = std::min(hi, lo + (uint32_t)(t + 1) * per);
        for (uint32_t e = lo + (uint32_t)t * per; e < k1; ++e)
            if (p->h_vm[e].n_dev_ops) { p->h_vm[e].n_dev_ops = 0u; p->transient[e] = 1; }
    };
    Workers::get().each(T, slice);
    p->lists_valid = false;
    if (p->d_prev_tr) p->attn_resync = true;      // p->transient moved without the device's copy
}

// Melange preamp: the default kernel re-factors the 12x12 system for every sample whose R_ldr moved, operation for operation like the
// reference (ow_melange_col.h).  OW_MEL_RANK1=1 selects the rank-one (Sherman-Morrison) kernel instead: mathematically the same, but
// without the LU's rounding noise, i.e. up to 1.8e-7 V away from the reference while R_ldr moves fast (DESIGN.md deviation 6);
// OW_MEL_LDS=1 the round-2 literal kernel (S of every engine in LDS); OW_MEL_GENERIC=1 the literal kernels without their fast path.
static inline bool eout_attention(const ow_pool* p, int ne) { return p->d_attn && (p->sw.eout_attn < 0 ? ne >= 8192 : p->sw.eout_attn != 0); }
static inline bool melange_rank_one(const ow_pool* p) { return p->sw.mel_rank1; }
static inline bool melange_lds_matrix(const ow_pool* p) { return p->sw.mel_lds; }
static inline bool melange_generic_only(const ow_pool* p) { return p->sw.mel_generic; }
// OW_MEL_ENG=1: lane = engine (ow_melange_eng.h: the rebuild once per engine instead of once per solver state, 64 engines per wavefront).
// Bit-identical; measured 39.2 against 40.4 ms per 131 072-engine block and 45 against 21 ms at 65 536 (one wavefront per SIMD): not the default.
static inline bool melange_lane_engine(const ow_pool* p, int) { return p->sw.mel_eng != 0; }

// Stages of the staged render.  Off by default: OW_PIPE=n (2..8) cuts big ranges (>= 32 768 engines) into n engine stages on their own
// streams, chained stage to stage, so that the output copy of a stage runs beside the kernels of the next one.  Measured: stages cost
// more than the copy overlap they buy ("what did not work").
static inline int pipeline_stages(const ow_pool* p, int ne) {
    if (ne < 32768) return 1;
    return p->sw.pipe ? p->sw.pipe : 1;
}
// OW_PIPE_OVERLAP=1: chain the stages voice kernel to voice kernel instead of stage to stage, so that the chain kernels of stage k run
// beside the voice kernel of stage k+1.  Measured slower at every stage count; kept as a switch so the measurement can be repeated.
static inline bool pipeline_overlap(const ow_pool* p) { return p->sw.pipe_overlap; }

// Deal the sounding voices of engines [e0, e0+ne) into wavefront-sized blocks (see ow_kernels.h, "Packed dispatch").
// general = engines whose status after the previous block reported a transient phase, or that receive ops in this block (a note-on
// starts an onset ramp and an attack-noise burst, a note-off a damper phase; nothing else starts one).
void build_voice_lists(ow_pool* p, int e0, int ne) {
    // Big ranges are cut into T engine slices that are packed independently (each slice starts on a block boundary, so at most
    // T - 1 blocks are less full than they could be): pass 1 sizes the three lists of every slice, a prefix sum places them,
    // pass 2 writes the entries.
    const int NP = pipeline_stages(p, ne);                     // stage boundaries exist whether or not this block uses them
    size_t T = ne >= 16384 ? std::min<size_t>(effective_cpus(), 32) : 1;
    if (NP > 1) T = std::max<size_t>(NP, T - T % (size_t)NP);   // stages are whole numbers of slices
    // slices (and with them the stages of a staged render) start on multiples of 32 engines: the chain kernels' workgroups then never
    // straddle a stage boundary, and per-workgroup scratch indexed by (first engine / 32 + block) is disjoint between stages
    const int per = (int)(((ne + T - 1) / T + 31) / 32 * 32);
    using Fill = ow_pool::SliceStart;
    Fill size[OW_MAX_SLICES];
    Fill* start = p->slice_start;                              // T <= 32; kept: stage k launches the blocks of its slices
    start[0] = Fill();
    p->slice_T = (int)T; p->slice_per = per;
    auto pack = [&](size_t t, uint32_t* S, uint32_t* G, uint32_t* Tl, uint32_t* A, Fill& f) {   // a null list: count (and hash) only
        auto pad = [](uint32_t* a, uint32_t& n) { while (n & 63u) { if (a) a[n] = 0xFFFFFFFFu; ++n; } };
        auto put_l = [&](int list, uint32_t* a, uint32_t& n, uint32_t e, uint64_t mask, bool own_block) {
            const uint32_t pc = (uint32_t)__builtin_popcountll(mask);
            if (own_block || (n & 63u) + pc > 64u) pad(a, n);
            if (a) for (uint64_t m = mask; m; m &= m - 1) a[n++] = (e << 6) | (uint32_t)__builtin_ctzll(m);
            else {
                n += pc;
                uint64_t* h = f.h[list];
                h[0] = (h[0] ^ ((uint64_t)e * 0x9E3779B97F4A7C15ull + mask)) * 0xFF51AFD7ED558CCDull;
                h[1] = (h[1] + mask * 0xC2B2AE3D27D4EB4Full + e) * 0x9FB21C651E98DF25ull + (h[1] >> 29);
            }
        };
        auto put = [&](uint32_t* a, uint32_t& n, uint32_t e, uint64_t mask, bool own_block) {
            put_l(&n == &f.s ? 0 : (&n == &f.g ? 1 : (&n == &f.t ? 2 : 3)), a, n, e, mask, own_block);
        };
        const int k1 = std::min(ne, (int)(t + 1) * per);
        for (int k = (int)t * per; k < k1; ++k) {
            const uint32_t e = (uint32_t)(e0 + k);
            const OwEngineArgs& a = p->h_args[e];
            if (a.main_mask) {
                if (p->sw.force_general) put(G, f.g, e, a.main_mask, false);
                else if (p->transient[e] || a.op_count) {
                    // in a transient phase, or about to be (ops pending).  A slot voice only damps while its slot is Releasing (the damper
                    // starts with the release, engine.rs:340-374): an engine without releasing slot voices is inside onset ramps / attack
                    // noise at most -- the attack variant of the steady kernel, 1.2-1.7 x its price instead of the general kernel's 2.6 x
                    const bool damping = (p->h_vm[e].st_mask[OW_VOICE_RELEASING] & a.main_mask) != 0ull;
                    if (!damping && p->sw.voice_attack) put(A, f.a, e, a.main_mask, false);
                    else put(G, f.g, e, a.main_mask, false);
                } else put(S, f.s, e, a.main_mask, false);
            }
            if (a.steal_mask) put(Tl, f.t, e, a.steal_mask, true);   // one engine per block: the crossfade early-out is per engine
        }
        pad(S, f.s); pad(G, f.g); pad(Tl, f.t); pad(A, f.a);
    };
    auto pass1 = [&](size_t t) { pack(t, nullptr, nullptr, nullptr, nullptr, size[t]); };
    Workers::get().each(T, pass1);
    for (size_t t = 0; t < T; ++t) { start[t + 1].s = start[t].s + size[t].s; start[t + 1].g = start[t].g + size[t].g; start[t + 1].t = start[t].t + size[t].t; start[t + 1].a = start[t].a + size[t].a; }
    const uint32_t fs = start[T].s, fg = start[T].g, fl = start[T].t, fa = start[T].a;
    // a list whose (engine, mask) sequence, slices and range are the ones its device copy was packed from is left alone: after a
    // whole-pool re-strike the attack and steal lists of the 128-sample sub-block serve the next block(s) as they are
    ow_pool::VoiceList* const vls[4] = {&p->vl_steady, &p->vl_general, &p->vl_steal, &p->vl_attack};
    const uint32_t fill[4] = {fs, fg, fl, fa};
    bool keep[4];
    for (int l = 0; l < 4; ++l) {
        uint64_t sig[3] = {0x243F6A8885A308D3ull ^ (uint64_t)T, 0x13198A2E03707344ull ^ (uint64_t)per, ((uint64_t)(uint32_t)e0 << 32) | (uint32_t)ne};
        for (size_t t = 0; t < T; ++t) {
            const uint32_t n_t = l == 0 ? size[t].s : (l == 1 ? size[t].g : (l == 2 ? size[t].t : size[t].a));
            sig[0] = (sig[0] * 0x100000001B3ull) ^ size[t].h[l][0] ^ ((uint64_t)n_t << 17);
            sig[1] = (sig[1] * 0xD6E8FEB86659FD93ull) + size[t].h[l][1] + n_t;
        }
        ow_pool::VoiceList& vl = *vls[l];
        keep[l] = fill[l] != 0 && vl.sig_valid && vl.n_blocks == fill[l] / 64 && vl.sig[0] == sig[0] && vl.sig[1] == sig[1] && vl.sig[2] == sig[2];
        // the signature only stands for the DEVICE copy once that copy has been enqueued (below): if anything throws in between, no list keeps
        // a signature whose upload never happened
        vl.sig[0] = sig[0]; vl.sig[1] = sig[1]; vl.sig[2] = sig[2]; vl.sig_valid = keep[l];
        vl.n_blocks = fill[l] / 64;
    }
    auto pass2 = [&](size_t t) {
        Fill f;   // slice-local counters: the slice regions start on block boundaries, so padding decisions match pass 1
        pack(t, keep[0] ? nullptr : p->vl_steady.h + start[t].s, keep[1] ? nullptr : p->vl_general.h + start[t].g,
             keep[2] ? nullptr : p->vl_steal.h + start[t].t, keep[3] ? nullptr : p->vl_attack.h + start[t].a, f);
    };
    if (!((keep[0] || !fs) && (keep[1] || !fg) && (keep[2] || !fl) && (keep[3] || !fa))) Workers::get().each(T, pass2);
    hipStream_t st = p->stream;
    for (int l = 0; l < 4; ++l)
        if (fill[l] && !keep[l]) {
            HIP_OK(hipMemcpyAsync(vls[l]->d, vls[l]->h, sizeof(uint32_t) * fill[l], hipMemcpyHostToDevice, st));
            vls[l]->sig_valid = true;
        }
    p->lists_e0 = e0; p->lists_ne = ne;
}

// one oscillator per phase group of the range (p->d_leaders, trem_leader_list)
static void launch_tremolo(ow_pool* p, hipStream_t tt, double* rbuf, int n_os) {
    const int I = (int)p->I, nl = p->n_lead;
    if (p->tremolo_kind == OW_TREMOLO_LEGACY_LFO) owdev::k_tremolo_lfo<<<dim3((nl + 63) / 64), dim3(64), 0, tt>>>(p->dK, p->d_cs, rbuf, I, (long long)n_os, p->d_leaders, nl, 0LL);
    else if (trem_wide(p, nl)) owdev::k_tremolo_wide<false><<<dim3((nl + 15) / 16), dim3(64), 0, tt>>>(p->dK, p->d_cs, rbuf, I, (long long)n_os, p->d_leaders, nl);
    else owdev::k_tremolo<<<dim3((nl + 63) / 64), dim3(64), 0, tt>>>(p->dK, p->d_cs, rbuf, I, n_os, p->d_leaders, nl);
}

// One render of `len` samples for engines [e0, e0+ne).  with_voices=false skips the voice kernels
// (warm-up of engines whose voices were just freed).
// out_host != nullptr: rows [e0, e0+ne) of the block are copied to out_host[(e - e0) * out_stride] as their stages finish.
void render_range(ow_pool* p, int e0, int ne, size_t len, bool with_voices, float* out_host = nullptr, size_t out_stride = 0) {
    const int I = (int)p->I;
    const int L = (int)len, Lcap = (int)p->Lcap;
    p->out_ld = len;
    hipStream_t st = p->stream, tt = p->stream_trem;
    vm_wait_download(p);                     // a burst applied on the device: the host's copy of the voice-pool states is complete from here on
    // ---- tremolo: CdS cell resistance of this block
    const int n_os = L * (p->hc.oversample ? 2 : 1);
    const size_t rb_half = (size_t)2 * p->Lcap * p->I;
    const bool chain = !p->voices_only;     // a voices-only pool (ow_render_note) stops at the voice sums
    const bool whole = e0 == 0 && ne == I;
    // (a) engines on the shared trajectory read r_ldr[t .. t + n_os) at their own t = clock - birth: make the store reach the oldest one's
    //     block (nothing to do unless this pool holds the process's oldest engine), plus one block ahead on the store's own stream
    owdev::OwTremSrc tsrc{nullptr, p->d_lead, nullptr, p->d_birth};
    hipEvent_t traj_ready = nullptr;
    if (chain && p->traj) {
        long long mn = p->min_birth;
        if (!whole) { mn = p->trem_clock; for (int k = 0; k < ne; ++k) if (p->h_birth[e0 + k] != OW_OFF_TRAJ) mn = std::min(mn, p->h_birth[e0 + k]); }
        if ((size_t)(p->trem_clock - mn) + (size_t)n_os > p->traj->cap_max) { trem_evict(p, e0, ne, n_os); mn = p->trem_clock; for (int k = 0; k < ne; ++k) if (p->h_birth[e0 + k] != OW_OFF_TRAJ) mn = std::min(mn, p->h_birth[e0 + k]); }
        size_t need = (size_t)(p->trem_clock - mn) + (size_t)n_os;
        bool ask = false, must = false, feed = false;
        {
            std::lock_guard<std::mutex> lk(p->traj->mu);
            must = need > p->traj->cap;                        // the helper did not get there in time (it is asked a lead + 30 s before)
            if (!must && p->traj->wants_growth(need)) { p->traj->grow_requested = true; ask = true; }
        }
        if (must) {                                            // cold: allocates on this thread, like the reference's buffer auto-grow
            try { p->traj->grow_to(std::max(need, p->traj->cap * 2)); }
            catch (const std::exception& ex) {
                // no memory for a longer store (a crowded device): it stays as long as it is -- as the helper thread's failure path decides --
                // and the engines that have outgrown it continue on oscillators of their own instead of retrying the allocation every block
                (void)hipGetLastError();
                { std::lock_guard<std::mutex> lk(p->traj->mu); p->traj->cap_max = p->traj->cap; p->traj->grow_requested = false; }
                std::fprintf(stderr, "openwurli-hip: tremolo trajectory store stays at %zu samples (%s)\n", p->traj->cap, ex.what());
                trem_evict(p, e0, ne, n_os);
                mn = p->trem_clock;
                for (int k = 0; k < ne; ++k) if (p->h_birth[e0 + k] != OW_OFF_TRAJ) mn = std::min(mn, p->h_birth[e0 + k]);
                need = (size_t)(p->trem_clock - mn) + (size_t)n_os;
            }
        }
        if (ask) traj_grower().ask(p->traj);
        {
            std::lock_guard<std::mutex> lk(p->traj->mu);
            traj_ready = p->traj->cover(need, (size_t)n_os, p->I < 4096 ? p->traj->lead : 0, &feed);
            tsrc.traj = p->traj->d_r + p->trem_clock;         // (under the lock: a growth swaps the buffer)
        }
        if (feed) traj_grower().feed(p->traj);
    }
    // (b) engines with an oscillator of their own (phase groups): already there if the block-ahead speculation hit
    const bool hit = chain && p->spec.valid && p->spec.e0 == e0 && p->spec.ne == ne && p->spec.n_os == n_os;
    if (p->spec.valid && !hit) {   // mis-speculated (different block length / engine range): roll the oscillator back
        HIP_OK(hipMemcpyAsync(p->d_cs, p->d_trem_backup, sizeof(double) * 18 * p->I, hipMemcpyDeviceToDevice, tt));
        p->spec.valid = false;
    }
    if (chain) {
        // a sub-range advances on its own: its engines leave the phase groups they share with engines outside it (no-op for the whole pool
        // and for a range that was just initialised); then the oscillators to run are the group leaders inside the range
        if (p->n_on_traj < p->I) trem_split_at_range(p, e0, ne);
        trem_leader_list(p, e0, ne);
        if (hit) {
            p->rb_cur ^= 1;            // the half the speculation filled
        } else if (p->n_lead > 0) {
            launch_tremolo(p, tt, p->d_rbuf + p->rb_cur * rb_half, n_os);
            HIP_OK(hipEventRecord(p->ev_trem[p->rb_cur], tt));
        }
    }
    const bool own_osc = chain && p->n_lead > 0;
    const double* rb_now = p->d_rbuf + p->rb_cur * rb_half;
    tsrc.rbuf = rb_now;
    const int rb_now_idx = p->rb_cur;
    // ---- next block, speculatively: back up the oscillator rows, then run ahead into the other half
    auto launch_block_ahead = [&] {
        const int nxt = p->rb_cur ^ 1;
        HIP_OK(hipMemcpyAsync(p->d_trem_backup, p->d_cs, sizeof(double) * 18 * p->I, hipMemcpyDeviceToDevice, tt));
        if (p->profiling) HIP_OK(hipEventRecord(p->ev[6], tt));
        launch_tremolo(p, tt, p->d_rbuf + nxt * rb_half, n_os);
        if (p->profiling) HIP_OK(hipEventRecord(p->ev[7], tt));
        HIP_OK(hipEventRecord(p->ev_trem[nxt], tt));
        p->spec.valid = true; p->spec.e0 = e0; p->spec.ne = ne; p->spec.n_os = n_os;
    };
    // The oscillators go first: they need nothing from the host, so they run while the host packs ops and voice lists, and their
    // wavefronts (one per SIMD at 65 536 of them) are resident before the voice kernel fills the rest.  (k_apply_ops is register-capped
    // so that it fits beside them.)
    if (own_osc) launch_block_ahead();
    else if (chain && p->profiling) { HIP_OK(hipEventRecord(p->ev[6], tt)); HIP_OK(hipEventRecord(p->ev[7], tt)); }
    // OW_TREM_SERIAL=1 (measurement switch, see trem_serialised): the voices of this block wait for the block-ahead oscillators
    if (own_osc && trem_serialised(p)) HIP_OK(hipStreamWaitEvent(st, p->ev_trem[p->rb_cur ^ 1], 0));
    // ---- per-engine args + ops: only engines whose host state changed are touched (the rest keep their
    // uploaded args; a steady-state step of a large pool does no per-engine host work here)
    // Large pools split the range over host threads: slice t counts its pending ops, a prefix sum places the slices in h_ops,
    // then every slice packs its own engines (a 65536-engine re-strike moves ~8 M ops; one thread took ~100 ms for it).
    // A steady block of the whole pool -- no engine touched since the last one -- skips the per-engine scans below (0.2 ms at 131 072)
    const bool untouched = whole && !__atomic_load_n(&p->dirty_any, __ATOMIC_RELAXED) && !p->args_stale && p->any_cache_valid;
    if (whole) __atomic_store_n(&p->dirty_any, (uint8_t)0, __ATOMIC_RELAXED);   // engines touched from here on belong to the next block
    size_t n_dirty = 0;
    if (!untouched) for (int k = 0; k < ne; ++k) n_dirty += p->dirty[e0 + k];     // dirty[] holds 0/1
    size_t T = (n_dirty >= 4096) ? std::min<size_t>(effective_cpus(), 32) : 1;
    const int per = (int)((ne + T - 1) / T);
    size_t cnt[OW_MAX_SLICES + 1] = {0};
    uint8_t dirty_t[OW_MAX_SLICES] = {0};
    auto count_slice = [&](size_t t) {
        size_t c = 0; uint8_t d = 0;
        const int k1 = std::min(ne, (int)(t + 1) * per);
        for (int k = (int)t * per; k < k1; ++k) {
            if (!p->dirty[e0 + k]) continue;
            d = 1;
            c += p->engines[e0 + k]->ops.size();
        }
        cnt[t + 1] = c; dirty_t[t] = d;
    };
    if (!untouched) Workers::get().each(T, count_slice);
    bool any_dirty = false;
    for (size_t t = 0; t < T; ++t) { any_dirty = any_dirty || dirty_t[t]; cnt[t + 1] += cnt[t]; }
    const size_t n_ops = cnt[T];
    ensure_ops_capacity(p, n_ops);
    if (any_dirty || p->args_stale) {
        auto pack_slice = [&](size_t t) {
            size_t op_pos = cnt[t];
            const int k1 = std::min(ne, (int)(t + 1) * per);
            for (int k = (int)t * per; k < k1; ++k) {
                OwEngineArgs& a = p->h_args[e0 + k];
                if (!p->dirty[e0 + k]) {
                    if (a.op_count || a.set_flags) { a.op_count = 0; a.set_flags = 0; }
                    continue;
                }
                ow_engine* en = p->engines[e0 + k];
                // engine.rs:471-473: a Free slot renders nothing unless it still carries a steal voice
                a.main_mask = en->vm->main_mask; a.steal_mask = en->vm->steal_mask;
                a.noise_on = en->noise_on ? 1u : 0u; a.thermal_gain = en->thermal_gain;
                a.pa_flags = en->rail_sag ? 1u : 0u;
                a.op_begin = (uint32_t)op_pos;
                a.op_count = (uint32_t)en->ops.size();
                if (!en->ops.empty()) std::memcpy(p->h_ops + op_pos, en->ops.data(), sizeof(OwOp) * en->ops.size());
                if (const uint32_t nd = en->vm->n_dev_ops) {
                    // the engine's queue was (also) written on the device (k_vm_events): it stays where it is; ops the host queued
                    // behind it are appended there after the upload (rare: single events between a burst and the render)
                    if (!en->ops.empty()) {
                        const uint32_t room = OW_VM_OPS_MAX - nd, nh = (uint32_t)std::min<size_t>(en->ops.size(), room);
                        if (nh < en->ops.size()) {
                            // Only with OW_MIDI_APPLY_EARLY=0 (a burst's queue left for this render), a burst that filled the engine's 192
                            // entries and single events behind it: the tail does not fit the device queue.  Counted and reported -- the
                            // default schedule applies a burst's queue inside ow_pool_midi and never gets here with nd != 0.
                            const uint64_t lost = g_ops_dropped.fetch_add(en->ops.size() - nh) + (en->ops.size() - nh);
                            if (lost == en->ops.size() - nh)
                                std::fprintf(stderr, "openwurli-hip: engine %zu: %zu queued slot ops do not fit behind a device-side MIDI burst (OW_MIDI_APPLY_EARLY=0) and are dropped\n",
                                             (size_t)(e0 + k), en->ops.size() - nh);
                        }
                        std::lock_guard<std::mutex> lk(p->op_tails_mu);
                        p->op_tails.push_back({(uint32_t)op_pos, (uint32_t)((e0 + k) * OW_VM_OPS_MAX) + nd, nh});
                        a.op_count = nd + nh;
                    } else a.op_count = nd;
                    a.op_begin = 0x80000000u | (uint32_t)((e0 + k) * OW_VM_OPS_MAX);
                    en->vm->n_dev_ops = 0;
                    if (!p->vm_host_dirty) p->vm_host_dirty = 1;
                }
                op_pos += en->ops.size();
                if (!en->ops.empty()) { en->ops.clear(); __atomic_fetch_sub(&p->host_ops_any, 1u, __ATOMIC_RELAXED); }
                a.set_flags = 0;
                if (en->depth.pending) { a.set_flags |= 1u; a.depth_target = en->depth.pending_value; en->depth.pending = false; }
                if (en->spk.pending)   { a.set_flags |= 2u; a.spk_target = en->spk.pending_value;     en->spk.pending = false; }
                if (en->volume.pending){ a.set_flags |= 4u; a.vol_target = en->volume.pending_value;  en->volume.pending = false; }
                p->dirty[e0 + k] = 0;
            }
        };
        Workers::get().each(T, pack_slice);
    }
    // One parallel pass over the args of the range (13 MB at 131 072 engines: ~1 ms per walk on one thread, and the blocks after a
    // whole-pool re-strike used to take two): does any engine sound, and which engines have ops (slice t lists its own into its part of
    // h_op_engines; the parts are closed up below).
    bool any_main = p->any_main_c, any_steal = p->any_steal_c;
    uint32_t n_act = 0;
    if (!untouched) {
        const size_t TA = ne >= 16384 ? std::min<size_t>(effective_cpus(), 32) : 1;
        const int per_a = (int)((ne + TA - 1) / TA);
        uint32_t act_n[OW_MAX_SLICES] = {0};
        uint8_t any_m[OW_MAX_SLICES] = {0}, any_s[OW_MAX_SLICES] = {0};
        auto scan_slice = [&](size_t t) {
            const int k0 = (int)t * per_a, k1 = std::min(ne, (int)(t + 1) * per_a);
            uint32_t n = 0; uint8_t m = 0, sl = 0;
            uint32_t* dst = p->h_op_engines + k0;
            for (int k = k0; k < k1; ++k) {
                const OwEngineArgs& a = p->h_args[e0 + k];
                m |= a.main_mask != 0; sl |= a.steal_mask != 0;
                if (a.op_count) dst[n++] = (uint32_t)(e0 + k);
            }
            act_n[t] = n; any_m[t] = m; any_s[t] = sl;
        };
        Workers::get().each(TA, scan_slice);
        any_main = false; any_steal = false;
        for (size_t t = 0; t < TA; ++t) {
            any_main |= any_m[t] != 0; any_steal |= any_s[t] != 0;
            if (act_n[t] && n_act != (uint32_t)(t * per_a)) std::memmove(p->h_op_engines + n_act, p->h_op_engines + t * per_a, sizeof(uint32_t) * act_n[t]);
            n_act += act_n[t];
        }
        if (whole) { p->any_main_c = any_main; p->any_steal_c = any_steal; p->any_cache_valid = true; }
    }
    // args carry one-shot fields (ops, setter targets): upload when anything changed, and once more afterwards to clear them
    if (any_dirty || p->args_stale) {
        HIP_OK(hipMemcpyAsync(p->d_args + e0, p->h_args + e0, sizeof(OwEngineArgs) * ne, hipMemcpyHostToDevice, st));
        p->args_stale = any_dirty;
    }
    HIP_OK(hipMemsetAsync(p->d_eout + e0, 0, sizeof(OwEngineOut) * ne, st));
    if (p->profiling) HIP_OK(hipEventRecord(p->ev[0], st));
    if (n_ops || p->dev_ops_pending) {
        if (n_ops) HIP_OK(hipMemcpyAsync(p->d_ops, p->h_ops, sizeof(OwOp) * n_ops, hipMemcpyHostToDevice, st));
        for (const ow_pool::OpTail& t : p->op_tails)
            HIP_OK(hipMemcpyAsync(p->d_ops_fix + t.dst, p->d_ops + t.src, sizeof(OwOp) * t.n, hipMemcpyDeviceToDevice, st));
        p->op_tails.clear();
        // one block per engine that has ops (listed by the scan above; an untouched range has none)
        if (n_act) {
            HIP_OK(hipMemcpyAsync(p->d_op_engines, p->h_op_engines, sizeof(uint32_t) * n_act, hipMemcpyHostToDevice, st));
            owdev::k_apply_ops<<<dim3(n_act), dim3(64), 0, st>>>(p->dK, p->d_nt, p->d_vrec, p->d_args, p->d_ops, p->d_op_engines, p->d_ops_fix);
        }
        if (whole) p->dev_ops_pending = false;
    }
    if (p->profiling) HIP_OK(hipEventRecord(p->ev[1], st));
    const bool voices = with_voices && (any_main || any_steal);
    if (voices) {
        // the lists depend on masks, pending ops (any_dirty) and the transient flags of the previous block (post_render_host)
        if (!p->lists_valid || any_dirty || p->lists_e0 != e0 || p->lists_ne != ne) build_voice_lists(p, e0, ne);
        p->lists_valid = !any_dirty;     // engines with ops were classified "general" for this block only
    }
    // ---- stages (see ow_pool::slice_start).  Without voices the lists are not built: one stage.
    const int NP = voices ? std::min(pipeline_stages(p, ne), p->slice_T) : 1;
    const bool overlap = pipeline_overlap(p);
    p->last_np = NP;
    bool steady_launched = false;
    if (voices && p->vl_steady.n_blocks) HIP_OK(hipMemsetAsync(p->d_skew_seen, 0, sizeof(uint32_t), st));
    if (NP > 1) HIP_OK(hipEventRecord(p->ev_ready, st));      // args, ops and voice lists are in place
    // A target inside a pinned block of ow_host_alloc is written by the output stage itself (it is mapped into the device's address space)
    float* out_direct = nullptr;
    if (out_host && p->sw.out_direct != 0 && out_stride >= len && ne > 0)
        out_direct = (float*)host_block_device_ptr(out_host, sizeof(float) * ((size_t)(ne - 1) * out_stride + len));
    for (int k = 0; k < NP; ++k) {
        bool direct_done = false;
        hipStream_t s = p->pipe_stream[k];                    // [0] == st
        const int t0 = k * p->slice_T / NP, t1 = (k + 1) * p->slice_T / NP;
        const int se0 = NP == 1 ? e0 : e0 + std::min(ne, t0 * p->slice_per);
        const int se1 = NP == 1 ? e0 + ne : e0 + std::min(ne, t1 * p->slice_per);
        const int sne = se1 - se0;
        if (k > 0) {
            HIP_OK(hipStreamWaitEvent(s, p->ev_ready, 0));
            HIP_OK(hipStreamWaitEvent(s, p->ev_voice_done[k - 1], 0));   // the kernels of the stages run one stage after the other
        }
        if (p->profiling) HIP_OK(hipEventRecord(p->ev_stage[k][0], s));
        if (voices) {
            const ow_pool::SliceStart& a0 = p->slice_start[NP == 1 ? 0 : t0];
            const ow_pool::SliceStart& a1 = p->slice_start[NP == 1 ? p->slice_T : t1];
            const unsigned bs = (a1.s - a0.s) / 64, bg = (a1.g - a0.g) / 64, bt = (a1.t - a0.t) / 64, ba = (a1.a - a0.a) / 64;
            if (ba) owdev::k_voice_steady<false, 1><<<dim3(ba), dim3(64), 0, s>>>(p->dK, p->d_vrec, p->vl_attack.d + a0.a, p->d_sum, p->d_eout, I, L, Lcap, nullptr);
            if (bs)
                {
                // voices on more than one jitter grid in some wavefront of the previous steady launch: the skewed variant (same samples)
                if (p->sw.voice_skew && p->skew_next)
                    owdev::k_voice_steady<true><<<dim3(bs), dim3(64), 0, s>>>(p->dK, p->d_vrec, p->vl_steady.d + a0.s, p->d_sum, p->d_eout, I, L, Lcap, p->d_skew_seen);
                else
                    owdev::k_voice_steady<false><<<dim3(bs), dim3(64), 0, s>>>(p->dK, p->d_vrec, p->vl_steady.d + a0.s, p->d_sum, p->d_eout, I, L, Lcap, p->d_skew_seen);
                steady_launched = true;
            }
            if (bg) {
                // the release variant of the steady kernel takes the blocks whose voices are all past onset and noise (each block decides by
                // itself, voice_steal_takes: what is left in this list are engines with a damping voice); k_voice (pass | 4) renders the others
                const bool rel = p->sw.voice_release && !p->sw.force_general;      // (force_general measures k_voice itself)
                if (rel) owdev::k_voice_steady<false, 3><<<dim3(bg), dim3(64), 0, s>>>(p->dK, p->d_vrec, p->vl_general.d + a0.g, p->d_sum, p->d_eout, I, L, Lcap, nullptr);
                owdev::k_voice<<<dim3(bg), dim3(64), 0, s>>>(p->dK, p->d_vrec, p->vl_general.d + a0.g, p->d_sum, p->d_eout, I, L, Lcap, rel ? 4 : 0);
            }
            if (bt) {
                // the steal variant of the steady kernel takes the engines whose steal voices are past onset and noise (each block decides
                // by itself, voice_steal_takes); k_voice (pass | 4) renders the others
                if (p->sw.voice_steal) owdev::k_voice_steady<false, 2><<<dim3(bt), dim3(64), 0, s>>>(p->dK, p->d_vrec, p->vl_steal.d + a0.t, p->d_sum, p->d_eout, I, L, Lcap, nullptr);
                owdev::k_voice<<<dim3(bt), dim3(64), 0, s>>>(p->dK, p->d_vrec, p->vl_steal.d + a0.t, p->d_sum, p->d_eout, I, L, Lcap, p->sw.voice_steal ? 5 : 1);
            }
        }
        if (overlap && k + 1 < NP) HIP_OK(hipEventRecord(p->ev_voice_done[k], s));
        if (p->profiling) HIP_OK(hipEventRecord(p->ev_stage[k][1], s));
        if (own_osc || hit) HIP_OK(hipStreamWaitEvent(s, p->ev_trem[rb_now_idx], 0));
        if (traj_ready) HIP_OK(hipStreamWaitEvent(s, traj_ready, 0));
        if (p->profiling) HIP_OK(hipEventRecord(p->ev_stage[k][2], s));
        const bool fused = chain && sne > 0 && chain_fused(p, sne);
        // k_chain_stream (ow_chain_stream.h): legacy preamp + behavioural amp, oversampled chain, pools too big for the quad kernels.
        // Default: when the block goes to a pinned host block (the point of it: no copy trails the launch); OW_CHAIN_STREAM=1 always.
        const bool streamed = chain && !fused && sne > 0 && p->hc.oversample && p->hc.preamp_kind == OW_PREAMP_LEGACY8 && p->power_amp_kind == OW_POWER_AMP_BEHAVIORAL &&
                              !preamp_wide(p, sne) && (p->sw.chain_stream < 0 ? out_direct != nullptr : p->sw.chain_stream == 1);
        if (fused && chain_row(p, sne)) {      // ... with one solver state per row of sixteen lanes (ow_chain_row.h): four preamp wavefronts + the output-stage one per eight engines
            if (p->hc.oversample)
                owdev::k_chain_row<true><<<dim3((sne + 7) / 8), dim3(320), 0, s>>>(p->dK, p->d_cs, p->d_args, p->d_eout, p->d_sum, tsrc, p->d_pre, p->d_out, I, L, Lcap, L, se0, sne);
            else
                owdev::k_chain_row<false><<<dim3((sne + 7) / 8), dim3(320), 0, s>>>(p->dK, p->d_cs, p->d_args, p->d_eout, p->d_sum, tsrc, p->d_pre, p->d_out, I, L, Lcap, L, se0, sne);
        } else if (fused) {    // small pool: preamp and output stage as two wavefronts of one workgroup (ow_chain_wide.h)
            if (p->hc.oversample)
                owdev::k_chain_fused<true><<<dim3((sne + 7) / 8), dim3(128), 0, s>>>(p->dK, p->d_cs, p->d_args, p->d_eout, p->d_sum, tsrc, p->d_pre, p->d_out, I, L, Lcap, L, se0, sne);
            else
                owdev::k_chain_fused<false><<<dim3((sne + 7) / 8), dim3(128), 0, s>>>(p->dK, p->d_cs, p->d_args, p->d_eout, p->d_sum, tsrc, p->d_pre, p->d_out, I, L, Lcap, L, se0, sne);
        } else if (sne > 0 && chain && streamed) {   // big oversampled pool: preamp and output stage alternate per 64-sample chunk in one launch, rows stored
            float* o2 = out_direct ? out_direct + (size_t)(se0 - e0) * out_stride : nullptr;      // straight into the caller's pinned block
            owdev::k_chain_stream<<<dim3((sne + 31) / 32), dim3(64), 0, s>>>(p->dK, p->d_cs, p->d_args, p->d_eout, p->d_sum, tsrc, p->d_pre, p->d_out, I, L, Lcap, L, se0, sne, o2, out_stride);
            direct_done = o2 != nullptr;
        } else if (sne > 0 && chain) {
            if (p->hc.preamp_kind == OW_PREAMP_MELANGE12 && !melange_rank_one(p) && p->hc.ml_sparse_ok && !melange_lds_matrix(p) && melange_lane_engine(p, sne))
                owdev::k_preamp_mel_eng<<<dim3((sne + 63) / 64), dim3(64), 0, s>>>(p->dK, p->d_cs, p->d_mel_settled, p->d_args, p->d_eout, p->d_sum, tsrc,
                                                                                 p->d_pre, p->d_noise, I, L, Lcap, se0, sne, melange_generic_only(p) ? 1 : 0,
                                                                                 p->d_mel_lu, p->mel_lu_ld);
            else if (p->hc.preamp_kind == OW_PREAMP_MELANGE12 && !melange_rank_one(p) && p->hc.ml_sparse_ok && !melange_lds_matrix(p))
                owdev::k_preamp_mel_col<<<dim3((sne + 31) / 32), dim3(64), 0, s>>>(p->dK, p->d_cs, p->d_mel_settled, p->d_args, p->d_eout, p->d_sum, tsrc,
                                                                                 p->d_pre, p->d_noise, I, L, Lcap, se0, sne, melange_generic_only(p) ? 1 : 0,
                                                                                 p->d_mel_lu, p->mel_lu_ld);
            else if (p->hc.preamp_kind == OW_PREAMP_MELANGE12 && !melange_rank_one(p))
                owdev::k_preamp_mel_lit<<<dim3((sne + 31) / 32), dim3(64), 0, s>>>(p->dK, p->d_cs, p->d_mel_settled, p->d_args, p->d_eout, p->d_sum, tsrc,
                                                                                 p->d_pre, p->d_noise, I, L, Lcap, se0, sne, melange_generic_only(p) ? 1 : 0, p->d_mel_lu);
            else if (p->hc.preamp_kind == OW_PREAMP_MELANGE12)
                owdev::k_preamp_mel<<<dim3((sne + 31) / 32), dim3(64), 0, s>>>(p->dK, p->d_cs, p->d_mel_settled, p->d_args, p->d_eout, p->d_sum, tsrc,
                                                                             p->d_pre, p->d_noise, I, L, Lcap, se0, sne);
            else if (preamp_wide(p, sne))
                owdev::k_preamp_wide<<<dim3((sne + 7) / 8), dim3(64), 0, s>>>(p->dK, p->d_cs, p->d_args, p->d_eout, p->d_sum, tsrc, p->d_pre, I, L, Lcap, se0, sne);
            else
                owdev::k_preamp<<<dim3((sne + 31) / 32), dim3(64), 0, s>>>(p->dK, p->d_cs, p->d_args, p->d_eout, p->d_sum, tsrc, p->d_pre, I, L, Lcap, se0, sne);
        }
        if (p->profiling) HIP_OK(hipEventRecord(p->ev_stage[k][3], s));
        if (!chain || fused || streamed) {
            // voice sums only / the output stage ran inside k_chain_fused or k_chain_stream
        } else if (sne > 0 && p->power_amp_kind == OW_POWER_AMP_MELANGE) {
            // more engines than one workgroup: dispatch them by falling demand of their last block (see k_post_mpa)
            // -- when the block has more engines than the chip holds at once (two workgroups of 32 per CU); below that every wavefront
            // is resident from the start, the block lasts as long as its slowest engine and the order cannot matter
            const bool ordered = power_amp_ordered(p) && ne > power_amp_resident_engines(p);
            if (ordered) {
                uint32_t* hist = p->d_pa_hist + (size_t)k * PA_ORDER_CLASSES;
                const uint32_t total = (uint32_t)L * (p->hc.oversample ? 2u : 1u);
                HIP_OK(hipMemsetAsync(hist, 0, sizeof(uint32_t) * PA_ORDER_CLASSES, s));
                owdev::k_pa_order_hist<<<dim3((sne + 255) / 256), dim3(256), 0, s>>>(p->d_pa_demand, se0, sne, total, hist);
                owdev::k_pa_order_scan<<<dim3(1), dim3(PA_ORDER_CLASSES), 0, s>>>(hist);
                owdev::k_pa_order_scatter<<<dim3((sne + 255) / 256), dim3(256), 0, s>>>(p->d_pa_demand, se0, sne, total, hist, p->d_pa_order);
            }
            owdev::k_post_mpa<<<dim3((sne + PA_EPB - 1) / PA_EPB), dim3(PA_WPB * 64), 0, s>>>(p->dK, p->dPa, p->d_pa_settled, p->d_cs, p->d_pa, p->d_args, p->d_eout, p->d_pre, p->d_out,
                                                                          p->d_pa_tap, I, L, L, se0, sne, ordered ? p->d_pa_order : nullptr, p->d_pa_demand);
        } else if (sne > 0) {
            float* o2 = out_direct ? out_direct + (size_t)(se0 - e0) * out_stride : nullptr;
            if (p->hc.oversample)
                owdev::k_post<true><<<dim3((sne + 31) / 32), dim3(64), 0, s>>>(p->dK, p->d_cs, p->d_args, p->d_eout, p->d_pre, p->d_out, I, L, L, se0, sne, o2, out_stride);
            else
                owdev::k_post<false><<<dim3((sne + 63) / 64), dim3(64), 0, s>>>(p->dK, p->d_cs, p->d_args, p->d_eout, p->d_pre, p->d_out, I, L, L, se0, sne, o2, out_stride);
            direct_done = o2 != nullptr;
        }
        if (p->profiling) HIP_OK(hipEventRecord(p->ev_stage[k][4], s));
        if (!overlap && k + 1 < NP) HIP_OK(hipEventRecord(p->ev_voice_done[k], s));   // the next stage computes while this one's rows are copied
        if (out_host && sne > 0 && !direct_done) {        // the stage's rows go out while the later stages still compute
            float* dst = out_host + (size_t)(se0 - e0) * out_stride;
            const float* src = p->d_out + (size_t)se0 * len;
            if (out_stride == len) HIP_OK(hipMemcpyAsync(dst, src, sizeof(float) * len * (size_t)sne, hipMemcpyDeviceToHost, s));
            else HIP_OK(hipMemcpy2DAsync(dst, out_stride * sizeof(float), src, len * sizeof(float), len * sizeof(float), (size_t)sne, hipMemcpyDeviceToHost, s));
        }
        if (k > 0) HIP_OK(hipEventRecord(p->ev_stage_done[k], s));
    }
    for (int k = 1; k < NP; ++k) HIP_OK(hipStreamWaitEvent(st, p->ev_stage_done[k], 0));
    if (chain && p->traj) {          // the engines of the range are n_os samples further along the trajectory
        if (whole) p->trem_clock += n_os;
        else {
            for (int k = 0; k < ne; ++k) if (p->h_birth[e0 + k] != OW_OFF_TRAJ) { p->h_birth[e0 + k] -= n_os; p->min_birth = std::min(p->min_birth, p->h_birth[e0 + k]); }
            owdev::k_trem_birth_shift<<<dim3((ne + 255) / 256), dim3(256), 0, st>>>(p->d_birth, e0, ne, -(long long)n_os);
        }
    }
    p->last_n_os = n_os;
    HIP_OK(hipGetLastError());
    if (steady_launched) {
        HIP_OK(hipMemcpyAsync(p->h_skew_seen, p->d_skew_seen, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        p->skew_pending = true;
    }
    if (eout_attention(p, ne)) {
        if (p->attn_resync) {
            std::memcpy(p->h_prev_tr, p->transient.data(), p->I);
            HIP_OK(hipMemcpyAsync(p->d_prev_tr, p->h_prev_tr, p->I, hipMemcpyHostToDevice, st));
            p->attn_resync = false;
        }
        owdev::k_eout_attention<<<dim3((ne + 255) / 256), dim3(256), 0, st>>>(p->d_eout, p->d_args, p->d_prev_tr, e0, ne, p->d_attn);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpyAsync(p->h_attn, p->d_attn, sizeof(uint64_t) * ((ne + 63) / 64), hipMemcpyDeviceToHost, st));
        p->attn_pending = true;
    } else {
        HIP_OK(hipMemcpyAsync(p->h_eout + e0, p->d_eout + e0, sizeof(OwEngineOut) * ne, hipMemcpyDeviceToHost, st));
        p->attn_pending = false;
    }
}

// host bookkeeping of ONE engine after a block: steal-fade countdown (engine.rs:490-493), NaN-guard frees (engine.rs:499-521,
// culprits identified in the same pass) and cleanup_voices (engine.rs:592-602)
// slot.steal_fade.saturating_sub(len) and the drop of a steal voice whose crossfade has ended (engine.rs:490-493): needs nothing from the
// block's status, so a whole-pool render does it for every engine while the kernels run (steal_countdown_early) instead of after them
void engine_steal_countdown(ow_engine* en, uint32_t l32) {
    OwVm& v = *en->vm;
    if (!v.steal_mask) return;
    vm_host_changed(en);
    for (uint64_t m = v.steal_mask; m; m &= m - 1) {
        const int s = __builtin_ctzll(m);
        v.steal_fade[s] = v.steal_fade[s] > l32 ? v.steal_fade[s] - l32 : 0u;
        if (v.steal_fade[s] == 0) { v.has_steal &= ~(1ull << s); en->sync_masks(s); }
    }
}
void engine_post_render(ow_engine* en, uint32_t l32, const OwEngineOut& o, bool steal_counted = false) {
    OwVm& v = *en->vm;
    vm_host_changed(en);
    if (!steal_counted) engine_steal_countdown(en, l32);
    if (o.sum_nonfinite) {
        en->nan_guard_fires += 1;
        for (int s = 0; s < OW_MAX_VOICES; ++s) {
            const uint64_t b = 1ull << s;
            if ((o.bad_main >> s) & 1ull) { en->set_state(s, OW_VOICE_FREE); v.has_voice &= ~b; }
            if ((o.bad_steal >> s) & 1ull) { v.has_steal &= ~b; v.steal_fade[s] = 0; }
            en->sync_masks(s);
        }
    }
    if (o.out_nonfinite) en->output_nan_resets += 1;
    for (uint64_t m = o.silent_mask & v.main_mask; m; m &= m - 1) {
        const int s = __builtin_ctzll(m);
        if (en->state_of(s) != OW_VOICE_FREE && ((v.has_voice >> s) & 1ull)) { en->set_state(s, OW_VOICE_FREE); v.has_voice &= ~(1ull << s); en->sync_masks(s); }
    }
}

// Second render of the voice-sum NaN guard (engine.rs:496-521).  When a block's voice sum was non-finite the reference zeroes the sum and
// renders every voice the engine still has AGAIN, freeing the ones whose output is non-finite; the survivors have then advanced 2 * len
// samples.  engine_post_render has already freed the voices that were non-finite in the first pass and dropped the steal voices whose
// crossfade ended (the reference drops those at the end of the first pass, before the guard looks); here the survivors are stepped by
// another `len` samples (k_voice in guard mode: same stepping, nothing summed), voices that turn non-finite in that second pass are freed
// too, and the status the block leaves behind (silent voices, transient phases) is the status after the second pass.  Cold path.
void engine_guard_second_pass_result(ow_engine* en, const OwEngineOut& o) {
    OwVm& v = *en->vm;
    vm_host_changed(en);
    for (int s = 0; s < OW_MAX_VOICES; ++s) {
        const uint64_t b = 1ull << s;
        bool touched = false;
        if ((o.bad_main >> s) & 1ull) { en->set_state(s, OW_VOICE_FREE); v.has_voice &= ~b; touched = true; }
        if ((o.bad_steal >> s) & 1ull) { v.has_steal &= ~b; v.steal_fade[s] = 0; touched = true; }
        if (touched) en->sync_masks(s);
    }
    for (uint64_t m = o.silent_mask & v.main_mask; m; m &= m - 1) {     // cleanup_voices sees the twice-advanced voices (engine.rs:461)
        const int s = __builtin_ctzll(m);
        if (en->state_of(s) != OW_VOICE_FREE && ((v.has_voice >> s) & 1ull)) { en->set_state(s, OW_VOICE_FREE); v.has_voice &= ~(1ull << s); en->sync_masks(s); }
    }
}
void guard_second_pass(ow_pool* p, const uint32_t* engs, size_t n_eng, size_t len) {
    hipStream_t st = p->stream;
    uint32_t nm = 0, ns = 0;          // entries; one block per engine and list
    for (size_t i = 0; i < n_eng; ++i) {
        const ow_engine* en = p->engines[engs[i]];
        auto put = [&](uint32_t* a, uint32_t& n, uint64_t mask) {
            if (!mask) return;
            for (uint64_t m = mask; m; m &= m - 1) a[n++] = (engs[i] << 6) | (uint32_t)__builtin_ctzll(m);
            while (n & 63u) a[n++] = 0xFFFFFFFFu;
        };
        put(p->vl_general.h, nm, en->vm->main_mask);
        put(p->vl_steal.h, ns, en->vm->steal_mask);
    }
    p->lists_valid = false;           // the list buffers were borrowed
    p->vl_general.sig_valid = false; p->vl_steal.sig_valid = false;
    // the guarded engines' status blocks: ONE clear, ONE gather and ONE transfer for the whole set (a bad parameter broadcast to a big
    // pool can put every engine here: 131 072 engines used to mean 262 144 runtime calls on the audio thread).  engs is pinned
    // (h_op_engines) and d_op_engines is free between renders.
    const unsigned nb = (unsigned)((n_eng + 255) / 256);
    HIP_OK(hipMemcpyAsync(p->d_op_engines, engs, sizeof(uint32_t) * n_eng, hipMemcpyHostToDevice, st));
    owdev::k_eout_clear_list<<<dim3(nb), dim3(256), 0, st>>>(p->d_eout, p->d_op_engines, (int)n_eng);
    const int I = (int)p->I, L = (int)len, Lcap = (int)p->Lcap;
    if (nm) {
        HIP_OK(hipMemcpyAsync(p->vl_general.d, p->vl_general.h, sizeof(uint32_t) * nm, hipMemcpyHostToDevice, st));
        owdev::k_voice<<<dim3(nm / 64), dim3(64), 0, st>>>(p->dK, p->d_vrec, p->vl_general.d, p->d_sum, p->d_eout, I, L, Lcap, 2);
    }
    if (ns) {
        HIP_OK(hipMemcpyAsync(p->vl_steal.d, p->vl_steal.h, sizeof(uint32_t) * ns, hipMemcpyHostToDevice, st));
        owdev::k_voice<<<dim3(ns / 64), dim3(64), 0, st>>>(p->dK, p->d_vrec, p->vl_steal.d, p->d_sum, p->d_eout, I, L, Lcap, 3);
    }
    owdev::k_eout_gather_list<<<dim3(nb), dim3(256), 0, st>>>(p->d_eout, p->d_op_engines, (int)n_eng, p->d_eout_packed);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(p->h_eout_packed, p->d_eout_packed, sizeof(OwEngineOut) * n_eng, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (size_t i = 0; i < n_eng; ++i) {
        OwEngineOut& o = p->h_eout[engs[i]];
        o = p->h_eout_packed[i];
        engine_guard_second_pass_result(p->engines[engs[i]], o);
        p->transient[engs[i]] = o.transient != 0u;
        o.sum_nonfinite = 1u;         // the block's voice sum stays "zeroed by the guard" for ow_pool_read_voice_sum
    }
}

// host bookkeeping after the block has been rendered (needs h_eout; call after stream sync)
// The steal-fade countdown of a whole-pool block while its kernels run (ow_pool_render, between the launches and the stream sync).
// After a whole-pool re-strike that is 64 counters and masks per engine over 120 MB of state: 2-3 ms on sixteen threads, GPU idle, when
// done after the block.  post_render_host then skips it (p->steal_counted).
void steal_countdown_early(ow_pool* p, size_t len) {
    p->steal_counted = false;
    const int ne = (int)p->I;
    if (!p->any_steal_c || !p->any_cache_valid || ne < 16384) return;
    const uint32_t l32 = (uint32_t)std::min<size_t>(len, 0xFFFFFFFFull);
    const size_t T = std::min<size_t>(effective_cpus(), 32);
    const int per = (int)((ne + T - 1) / T);
    auto slice = [&](size_t t) {
        const int k1 = std::min(ne, (int)(t + 1) * per);
        for (int k = (int)t * per; k < k1; ++k)
            if (p->h_args[k].steal_mask) engine_steal_countdown(p->engines[k], l32);
    };
    Workers::get().each(T, slice);
    p->steal_counted = true;
}

void post_render_host(ow_pool* p, int e0, int ne, size_t len) {
    if (p->skew_pending) { p->skew_pending = false; p->skew_next = *p->h_skew_seen != 0u; }
    const uint32_t l32 = (uint32_t)std::min<size_t>(len, 0xFFFFFFFFull);
    const bool steal_counted = p->steal_counted;
    p->steal_counted = false;
    // what one engine's status block asks of the host; returns bit 0 = the voice lists changed, bit 1 = misdispatch, bit 2 = voice-sum guard
    auto one = [&](int e) -> uint8_t {
        const OwEngineOut& o = p->h_eout[e];
        const OwEngineArgs& a = p->h_args[e];
        uint8_t r = 0;
        if (o.transient == 2u) r |= 2;
        const uint8_t tr = o.transient != 0u;
        if (tr != p->transient[e]) { p->transient[e] = tr; r |= 1; }
        // fast path on the contiguous status/args arrays: nothing to book-keep for this engine
        if ((steal_counted || !a.steal_mask) && !(o.silent_mask & a.main_mask) && !o.sum_nonfinite && !o.out_nonfinite) return r;
        if (o.sum_nonfinite) r |= 4;
        engine_post_render(p->engines[e], l32, o, steal_counted);
        return r;
    };
    auto second_pass = [&] {   // voice-sum NaN guard fired somewhere: the reference's second render pass for those engines
        uint32_t* engs = p->h_op_engines;     // pinned scratch of I entries, free between renders
        size_t n = 0;
        for (int k = 0; k < ne; ++k) if (p->h_eout[e0 + k].sum_nonfinite) engs[n++] = (uint32_t)(e0 + k);
        guard_second_pass(p, engs, n, len);
    };
    if (p->attn_pending) {
        // the block left one bit per engine (k_eout_attention): nothing set = nothing to do, the usual case of a big pool
        p->attn_pending = false;
        const int words = (ne + 63) / 64;
        size_t count = 0;
        for (int w = 0; w < words; ++w) count += (size_t)__builtin_popcountll(p->h_attn[w]);
        if (p->eout_all_live) { std::memset(p->h_eout, 0, sizeof(OwEngineOut) * p->I); p->eout_all_live = false; }
        else for (uint32_t e : p->eout_live) std::memset(&p->h_eout[e], 0, sizeof(OwEngineOut));
        p->eout_live.clear();
        if (count == 0) return;
        HIP_OK(hipMemcpyAsync(p->h_eout + e0, p->d_eout + e0, sizeof(OwEngineOut) * ne, hipMemcpyDeviceToHost, p->stream));
        HIP_OK(hipStreamSynchronize(p->stream));
        if (count * 8 <= (size_t)ne) {
            uint8_t r = 0;
            for (int w = 0; w < words; ++w)
                for (uint64_t m = p->h_attn[w]; m; m &= m - 1) {
                    const int e = e0 + w * 64 + __builtin_ctzll(m);
                    r |= one(e);
                    p->eout_live.push_back((uint32_t)e);
                }
            if (r & 1) p->lists_valid = false;
            if (r & 2) set_err("voice dispatch: an engine in a transient phase was sent to the steady kernel");
            if (r & 4) { second_pass(); p->attn_resync = true; }
            return;
        }
        p->eout_all_live = true;               // many engines (a whole-pool re-strike): the sliced scan below
    } else if (p->d_attn) {
        p->eout_all_live = true;
    }
    // engines are independent: after a whole-pool re-strike every engine has 64 steal fades to count down and 64 masks to
    // update (40 ms on one thread for 65 536 engines), so large ranges are cut into slices like the MIDI and op packing are
    const size_t T = ne >= 16384 ? std::min<size_t>(effective_cpus(), 32) : 1;
    const int per = (int)((ne + T - 1) / T);
    uint8_t res[OW_MAX_SLICES] = {0};
    auto slice = [&](size_t t) {
        const int k1 = std::min(ne, (int)(t + 1) * per);
        uint8_t r = 0;
        for (int k = (int)t * per; k < k1; ++k) r |= one(e0 + k);
        res[t] = r;
    };
    // a steady block of a big pool has (almost) nothing to do per engine, which is not worth starting threads for (~0.1 ms): estimate
    // the engines with steal fades / silent voices from every 64th one and go parallel from ~8 000 of them
    long need = 0;
    if (T > 1)
        for (int k = 0; k < ne; k += 64)
            need += p->h_args[e0 + k].steal_mask != 0 || (p->h_eout[e0 + k].silent_mask & p->h_args[e0 + k].main_mask) != 0;
    const bool busy = need * 64 > 8192;
    if (T == 1 || !busy) {
        for (size_t t = 0; t < T; ++t) slice(t);
    } else {
        Workers::get().each(T, slice);
    }
    uint8_t r = 0;
    for (size_t t = 0; t < T; ++t) r |= res[t];
    if (r & 1) { p->lists_valid = false; if (p->d_prev_tr) p->attn_resync = true; }    // p->transient moved without the device's copy
    if (r & 2) set_err("voice dispatch: an engine in a transient phase was sent to the steady kernel");
    if (r & 4) { second_pass(); if (p->d_prev_tr) p->attn_resync = true; }
}

void collect_profile(ow_pool* p) {
    if (!p->profiling) return;
    if (p->voices_only) { p->last_ms[2] = 0.f; }
    else hipEventSynchronize(p->ev[7]);   // in a small pool the block-ahead tremolo outlasts the audio stream; profiling waits for it, a normal render does not
    hipEventElapsedTime(&p->last_ms[0], p->ev[0], p->ev[1]);   // ops
    if (!p->voices_only) hipEventElapsedTime(&p->last_ms[2], p->ev[6], p->ev[7]);   // tremolo (own stream)
    // per-stage intervals on the stage streams, summed: voices run one stage after the other (the sum is the voice kernels' time, with
    // the previous stage's chain kernels running beside them); preamp / post of a stage overlap the next stage's voices
    float v = 0.f, pr = 0.f, po = 0.f, x = 0.f;
    for (int k = 0; k < p->last_np; ++k) {
        hipEventElapsedTime(&x, p->ev_stage[k][0], p->ev_stage[k][1]); v += x;
        hipEventElapsedTime(&x, p->ev_stage[k][2], p->ev_stage[k][3]); pr += x;
        hipEventElapsedTime(&x, p->ev_stage[k][3], p->ev_stage[k][4]); po += x;
    }
    p->last_ms[1] = v; p->last_ms[3] = pr; p->last_ms[4] = po;
}

// WurliEngine::warm_up (engine.rs:261-270): 0.6 s of render() in 512-sample blocks
void warm_up_range(ow_pool* p, int e0, int ne) {
    const size_t total = (size_t)owhip::sat_u32(p->hc.sr * 0.6);
    size_t done = 0;
    while (done < total) {
        const size_t len = std::min<size_t>(512, total - done);
        render_range(p, e0, ne, len, true);
        HIP_OK(hipStreamSynchronize(p->stream));
        post_render_host(p, e0, ne, len);
        done += len;
    }
}

void engine_host_reset(ow_engine* en) {  // host half of WurliEngine::reset (engine.rs:231-244)
    vm_host_current(en); vm_host_changed(en);
    const uint8_t mlp = en->vm->mlp_enabled;
    vm_init(*en->vm);                          // every slot Free, no voices, age counter 0, sustain up; nothing queued on the device either
    en->vm->mlp_enabled = mlp;                 // (a parameter, not state: reset() does not touch it)
    en->touch();
    if (!en->ops.empty()) { en->ops.clear(); if (en->host_ops_any) __atomic_fetch_sub(en->host_ops_any, 1u, __ATOMIC_RELAXED); }
    // snap_to(target): a pending retarget would ramp; the device snaps current := target, so drop the ramp request
    en->volume.pending = en->depth.pending = en->spk.pending = false;
}

void push_op(ow_engine* en, uint8_t type, int slot, uint8_t note, bool mlp, uint32_t seed, double vel) { en->push(type, slot, note, mlp, seed, vel); }

ow_pool* pool_create(double sample_rate, size_t n_engines, int device, int preamp_kind, int power_amp_kind = OW_POWER_AMP_BEHAVIORAL,
                     int tremolo_kind = OW_TREMOLO_TWIN_T, bool voices_only = false, bool no_traj) {
    if (!(sample_rate > 0.0) || n_engines == 0) throw std::runtime_error("invalid sample rate or engine count");
    if (tremolo_kind != OW_TREMOLO_TWIN_T && tremolo_kind != OW_TREMOLO_LEGACY_LFO) throw std::runtime_error("unknown tremolo_kind");
    if (preamp_kind != OW_PREAMP_LEGACY8 && preamp_kind != OW_PREAMP_MELANGE12) throw std::runtime_error("unknown preamp_kind");
    if (power_amp_kind != OW_POWER_AMP_BEHAVIORAL && power_amp_kind != OW_POWER_AMP_MELANGE) throw std::runtime_error("unknown power_amp_kind");
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) throw std::runtime_error("no HIP device: openwurli-hip has no CPU fallback");
    HIP_OK(hipSetDevice(device));
    ow_pool* p = new ow_pool();
    p->device = device;
    p->I = n_engines;
    p->power_amp_kind = power_amp_kind;
    p->tremolo_kind = tremolo_kind;
    p->voices_only = voices_only;
    HIP_OK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&p->stream_trem, hipStreamNonBlocking));
    for (auto& e : p->ev) HIP_OK(hipEventCreate(&e));
    p->pipe_stream[0] = p->stream;
    for (int k = 1; k < OW_MAX_STAGES; ++k) HIP_OK(hipStreamCreateWithFlags(&p->pipe_stream[k], hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&p->ev_ready, hipEventDisableTiming));
    for (int k = 0; k < OW_MAX_STAGES; ++k) {
        HIP_OK(hipEventCreateWithFlags(&p->ev_voice_done[k], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&p->ev_stage_done[k], hipEventDisableTiming));
        for (auto& e : p->ev_stage[k]) HIP_OK(hipEventCreate(&e));
    }
    for (auto& e : p->ev_trem) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_OK(hipMalloc(&p->d_trem_backup, sizeof(double) * 18 * n_engines));
    HIP_OK(hipMalloc(&p->d_trem_settled, sizeof(double) * 18));
    HIP_OK(hipMalloc(&p->d_zero, sizeof(uint32_t)));
    HIP_OK(hipMemsetAsync(p->d_zero, 0, sizeof(uint32_t), p->stream));
    HIP_OK(hipMalloc(&p->d_birth, sizeof(long long) * n_engines));
    HIP_OK(hipMalloc(&p->d_evict, sizeof(unsigned long long) * 3 * n_engines));
    p->h_birth.assign(n_engines, OW_OFF_TRAJ);
    p->sw = Switches::from_env();
    if (no_traj) p->sw.trem_traj = false;
    HIP_OK(hipMalloc(&p->dK, sizeof(OwConsts)));
    HIP_OK(hipMalloc(&p->dK48, sizeof(OwConsts)));
    HIP_OK(hipMalloc(&p->d_nt, sizeof(double) * NT_COUNT * 64));
    HIP_OK(hipMalloc(&p->d_vrec, sizeof(double) * n_engines * 2 * OW_VREC_DOUBLES));
    HIP_OK(hipMalloc(&p->d_cs, sizeof(double) * CS_COUNT * n_engines));
    HIP_OK(hipMalloc(&p->d_args, sizeof(OwEngineArgs) * n_engines));
    HIP_OK(hipMalloc(&p->d_eout, sizeof(OwEngineOut) * n_engines));
    HIP_OK(hipHostMalloc(&p->h_args, sizeof(OwEngineArgs) * n_engines));
    HIP_OK(hipHostMalloc(&p->h_vm, sizeof(OwVm) * n_engines));
    HIP_OK(hipEventCreateWithFlags(&p->ev_vm, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&p->ev_vm_events, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&p->ev_vm_up, hipEventDisableTiming));
    HIP_OK(hipHostMalloc(&p->h_eout, sizeof(OwEngineOut) * n_engines));
    HIP_OK(hipMalloc(&p->d_eout_packed, sizeof(OwEngineOut) * n_engines));
    HIP_OK(hipHostMalloc(&p->h_eout_packed, sizeof(OwEngineOut) * n_engines));
    HIP_OK(hipMalloc(&p->d_skew_seen, sizeof(uint32_t)));
    HIP_OK(hipMemset(p->d_skew_seen, 0, sizeof(uint32_t)));
    HIP_OK(hipHostMalloc(&p->h_skew_seen, sizeof(uint32_t)));
    *p->h_skew_seen = 0u;
    if (n_engines >= 64) {      // status summary of big ranges (k_eout_attention)
        HIP_OK(hipMalloc(&p->d_attn, sizeof(uint64_t) * ((n_engines + 63) / 64)));
        HIP_OK(hipHostMalloc(&p->h_attn, sizeof(uint64_t) * ((n_engines + 63) / 64)));
        HIP_OK(hipMalloc(&p->d_prev_tr, n_engines));
        HIP_OK(hipMemset(p->d_prev_tr, 0, n_engines));
        HIP_OK(hipHostMalloc(&p->h_prev_tr, n_engines));
        p->eout_live.reserve(n_engines / 8 + 64);       // the summary path lists at most ne / 8 engines: nothing grows on the render path
    }
    for (ow_pool::VoiceList* vl : {&p->vl_steady, &p->vl_general, &p->vl_steal, &p->vl_attack}) {   // worst case: one block per engine
        HIP_OK(hipMalloc(&vl->d, sizeof(uint32_t) * 64 * n_engines));
        HIP_OK(hipHostMalloc(&vl->h, sizeof(uint32_t) * 64 * n_engines));
    }
    p->transient.assign(n_engines, 0);
    HIP_OK(hipMalloc(&p->d_lead, sizeof(uint32_t) * n_engines));
    HIP_OK(hipHostMalloc(&p->h_lead, sizeof(uint32_t) * n_engines));
    HIP_OK(hipMalloc(&p->d_leaders, sizeof(uint32_t) * n_engines));
    HIP_OK(hipHostMalloc(&p->h_leaders, sizeof(uint32_t) * n_engines));
    HIP_OK(hipMalloc(&p->d_copy, sizeof(uint32_t) * 2 * n_engines));
    HIP_OK(hipHostMalloc(&p->h_copy, sizeof(uint32_t) * 2 * n_engines));
    p->grp_in.assign(n_engines, 0); p->grp_out.assign(n_engines, 0);
    for (size_t i = 0; i < n_engines; ++i) p->h_lead[i] = (uint32_t)i;   // singletons until the chain state is replicated below
    HIP_OK(hipMalloc(&p->d_snap, sizeof(double) * 3 * n_engines));
    HIP_OK(hipHostMalloc(&p->h_snap, sizeof(double) * 3 * n_engines));
    ensure_ops_capacity(p, (size_t)3 * OW_MAX_VOICES * n_engines);   // a whole-keyboard re-strike of every engine: no allocation in render
    Workers::get();                                                  // start the host worker threads now, not inside the first render
    HIP_OK(hipMalloc(&p->d_op_engines, sizeof(uint32_t) * n_engines));
    HIP_OK(hipHostMalloc(&p->h_op_engines, sizeof(uint32_t) * n_engines));
    std::memset(p->h_args, 0, sizeof(OwEngineArgs) * n_engines);
    std::memset(p->h_eout, 0, sizeof(OwEngineOut) * n_engines);
    HIP_OK(hipMemsetAsync(p->d_vrec, 0, sizeof(double) * n_engines * 2 * OW_VREC_DOUBLES, p->stream));
    HIP_OK(hipMemsetAsync(p->d_cs, 0, sizeof(double) * CS_COUNT * n_engines, p->stream));
    alloc_stream_buffers(p, n_engines == 1 ? (size_t)OW_MAX_BLOCK : (size_t)1024);  // engine.rs:25 MAX_BLOCK_SIZE for a lone engine
    if (power_amp_kind == OW_POWER_AMP_MELANGE) {
        HIP_OK(hipMalloc(&p->dPa, sizeof(OwPaConsts)));
        HIP_OK(hipMalloc(&p->d_pa, sizeof(double) * owdev::PAS_COUNT * n_engines));
        HIP_OK(hipMemsetAsync(p->d_pa, 0, sizeof(double) * owdev::PAS_COUNT * n_engines, p->stream));
        HIP_OK(hipMalloc(&p->d_pa_settled, sizeof(double) * owdev::PAS_CIRCUIT_END));
        pa_settled_to_device(device, p->d_pa_settled, p->stream);
        HIP_OK(hipMalloc(&p->d_pa_demand, sizeof(uint32_t) * n_engines));
        HIP_OK(hipMemsetAsync(p->d_pa_demand, 0, sizeof(uint32_t) * n_engines, p->stream));
        HIP_OK(hipMalloc(&p->d_pa_order, sizeof(uint32_t) * n_engines));
        HIP_OK(hipMalloc(&p->d_pa_hist, sizeof(uint32_t) * PA_ORDER_CLASSES * OW_MAX_STAGES));
    }
    upload_consts(p, sample_rate, preamp_kind);
    owdev::k_note_table<<<dim3(1), dim3(64), 0, p->stream>>>(p->d_nt);
    if (preamp_kind == OW_PREAMP_MELANGE12) {
        HIP_OK(hipMalloc(&p->d_mel_settled, sizeof(double) * 18));
        // LU workspace of the generic rebuild (the fallback of both literal kernels): the column-streamed kernel wants one [144] column
        // per lane = (engine, state), lane-minor, + 32 spare engines for the masked lanes of a last partial wavefront; the LDS-matrix
        // kernel one [144][32] slab per workgroup
        p->mel_lu_ld = 2 * (n_engines + 32);
        HIP_OK(hipMalloc(&p->d_mel_lu, sizeof(double) * 144 * std::max<size_t>(p->mel_lu_ld, 32 * ((size_t)(n_engines + 31) / 32 + OW_MAX_SLICES + 1))));
        mel_settled_to_device(device, p->d_mel_settled, p->stream);
        // Noise streams: the reference clones one process-wide state whose RNGs were seeded from the clock (master seed 0,
        // gen_preamp.rs:1512-1521 via melange_adapter.rs:12-29), so every engine of a process starts on the same streams.
        HIP_OK(hipMalloc(&p->d_noise, sizeof(double) * NZ_COUNT * n_engines));
        HIP_OK(hipMemsetAsync(p->d_noise, 0, sizeof(double) * NZ_COUNT * n_engines, p->stream));
        std::vector<uint64_t> seeds(n_engines, process_noise_seed());
        HIP_OK(hipMemcpyAsync(p->d_noise + (size_t)NZ_SEED * n_engines, seeds.data(), sizeof(uint64_t) * n_engines, hipMemcpyHostToDevice, p->stream));
        HIP_OK(hipStreamSynchronize(p->stream));
    }
    p->engines.resize(n_engines);
    p->dirty.assign(n_engines, 1);
    for (size_t i = 0; i < n_engines; ++i) {
        ow_engine* en = new ow_engine();
        en->pool = p;
        en->index = i;
        en->dirty = &p->dirty[i]; en->dirty_any = &p->dirty_any;
        en->vm = &p->h_vm[i];
        vm_init(*en->vm);
        en->host_ops_any = &p->host_ops_any;
        en->sr = sample_rate;
        // Room for a whole-keyboard re-strike (damper + move-to-steal + note-on per key) from the start: growing 65 536 op lists
        // from 64 to 192 entries inside the first re-strike cost 180 ms of reallocation and page faults on the MIDI threads
        // (later ones take 8 ms).  Same reason as ensure_buffer_capacity: no allocation where the events arrive.
        en->ops.reserve(3 * OW_MAX_VOICES);
        p->engines[i] = en;
    }
    if (voices_only) {          // Voice::render_note has no chain (voice.rs:191-221): nothing to initialise, nothing to settle
        for (size_t i = 0; i < n_engines; ++i) p->h_lead[i] = 0u;
        HIP_OK(hipStreamSynchronize(p->stream));
        return p;
    }
    // WurliEngine::new for engine 0 on the device, then replicate (every engine of a fresh pool is identical)
    chain_init_range(p, 0, 1, INIT_NEW, std::vector<double>(1, 0.5));
    if (n_engines > 1)
        owdev::k_chain_replicate<<<dim3((unsigned)((n_engines + 63) / 64)), dim3(64), 0, p->stream>>>(p->d_cs, (int)n_engines, 0, 0, (int)n_engines);
    if (power_amp_kind == OW_POWER_AMP_MELANGE && n_engines > 1)          // the amp state is not part of the replicated chain rows
        owdev::k_mpa_init<<<dim3((unsigned)((n_engines + 63) / 64)), dim3(64), 0, p->stream>>>(p->dPa, p->d_pa_settled, p->d_pa, (int)n_engines, 0, (int)n_engines, 1);
    if (p->traj) {                    // every engine at t = 0 of the shared trajectory
        for (size_t i = 0; i < n_engines; ++i) { p->h_birth[i] = p->trem_clock; p->h_lead[i] = (uint32_t)i; }
        traj_upload_births(p, 0, (int)n_engines);
        traj_recount(p);
    } else {
        for (size_t i = 0; i < n_engines; ++i) p->h_lead[i] = 0u;         // identical oscillators: one tremolo phase group led by engine 0
    }
    trem_groups_changed(p);
    if (p->d_noise)   // the replicated chain state does not carry the noise columns: seed every engine's streams
        owdev::k_mel_noise_seed<<<dim3((unsigned)((n_engines + 63) / 64)), dim3(64), 0, p->stream>>>(p->d_noise, (int)n_engines, 0, (int)n_engines);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(p->stream));
    return p;
}

void pool_destroy(ow_pool* p) {
    if (!p) return;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    free_stream_buffers(p);
    hipFree(p->dK); hipFree(p->dK48); hipFree(p->d_nt); hipFree(p->d_vrec); hipFree(p->d_cs);
    if (p->d_mel_settled) hipFree(p->d_mel_settled);
    if (p->d_mel_lu) hipFree(p->d_mel_lu);
    if (p->d_noise) hipFree(p->d_noise);
    if (p->dPa) hipFree(p->dPa);
    if (p->d_pa) hipFree(p->d_pa);
    if (p->d_pa_settled) hipFree(p->d_pa_settled);
    if (p->d_pa_tap) hipFree(p->d_pa_tap);
    if (p->d_pa_demand) hipFree(p->d_pa_demand);
    if (p->d_pa_order) hipFree(p->d_pa_order);
    if (p->d_pa_hist) hipFree(p->d_pa_hist);
    hipFree(p->d_args); hipFree(p->d_eout);
    if (p->d_ops) hipFree(p->d_ops);
    if (p->h_ops) hipHostFree(p->h_ops);
    for (ow_pool::VoiceList* vl : {&p->vl_steady, &p->vl_general, &p->vl_steal, &p->vl_attack}) { if (vl->d) hipFree(vl->d); if (vl->h) hipHostFree(vl->h); }
    if (p->d_op_engines) hipFree(p->d_op_engines);
    if (p->h_op_engines) hipHostFree(p->h_op_engines);
    if (p->d_lead) hipFree(p->d_lead);
    if (p->h_lead) hipHostFree(p->h_lead);
    if (p->d_leaders) hipFree(p->d_leaders);
    if (p->h_leaders) hipHostFree(p->h_leaders);
    if (p->d_copy) hipFree(p->d_copy);
    if (p->h_copy) hipHostFree(p->h_copy);
    if (p->d_snap) hipFree(p->d_snap);
    if (p->h_snap) hipHostFree(p->h_snap);
    hipHostFree(p->h_args); hipHostFree(p->h_eout);
    if (p->h_vm) hipHostFree(p->h_vm);
    if (p->d_vm) hipFree(p->d_vm);
    if (p->d_ops_fix) hipFree(p->d_ops_fix);
    if (p->h_ev) hipHostFree(p->h_ev);
    if (p->d_ev) hipFree(p->d_ev);
    if (p->d_ev_begin) hipFree(p->d_ev_begin);
    if (p->d_vm_ovf) hipFree(p->d_vm_ovf);
    if (p->h_vm_ovf) hipHostFree(p->h_vm_ovf);
    if (p->ev_vm) hipEventDestroy(p->ev_vm);
    if (p->ev_vm_events) hipEventDestroy(p->ev_vm_events);
    if (p->ev_vm_up) hipEventDestroy(p->ev_vm_up);
    if (p->d_eout_packed) hipFree(p->d_eout_packed);
    if (p->h_eout_packed) hipHostFree(p->h_eout_packed);
    if (p->d_skew_seen) hipFree(p->d_skew_seen);
    if (p->h_skew_seen) hipHostFree(p->h_skew_seen);
    if (p->d_attn) hipFree(p->d_attn);
    if (p->h_attn) hipHostFree(p->h_attn);
    if (p->d_prev_tr) hipFree(p->d_prev_tr);
    if (p->h_prev_tr) hipHostFree(p->h_prev_tr);
    for (auto& e : p->ev) if (e) hipEventDestroy(e);
    for (int k = 1; k < OW_MAX_STAGES; ++k) if (p->pipe_stream[k]) { hipStreamSynchronize(p->pipe_stream[k]); hipStreamDestroy(p->pipe_stream[k]); }
    if (p->ev_ready) hipEventDestroy(p->ev_ready);
    for (int k = 0; k < OW_MAX_STAGES; ++k) {
        if (p->ev_voice_done[k]) hipEventDestroy(p->ev_voice_done[k]);
        if (p->ev_stage_done[k]) hipEventDestroy(p->ev_stage_done[k]);
        for (auto& e : p->ev_stage[k]) if (e) hipEventDestroy(e);
    }
    for (auto& e : p->ev_trem) if (e) hipEventDestroy(e);
    if (p->d_trem_backup) hipFree(p->d_trem_backup);
    if (p->d_trem_settled) hipFree(p->d_trem_settled);
    if (p->d_zero) hipFree(p->d_zero);
    if (p->d_birth) hipFree(p->d_birth);
    if (p->d_evict) hipFree(p->d_evict);
    if (p->stream_trem) hipStreamDestroy(p->stream_trem);
    if (p->stream) hipStreamDestroy(p->stream);
    for (ow_engine* en : p->engines) delete en;
    delete p;
}

template <typename F>
bool guarded(const char* what, F&& f) {  // realtime entry points never fail: record the error, degrade; false = it failed
    try { f(); return true; }
    catch (const std::exception& ex) { set_err(std::string(what) + ": " + ex.what()); std::fprintf(stderr, "openwurli-hip: %s: %s\n", what, ex.what()); return false; }
}

}  // namespace

extern "C" {

int ow_abi_version(void) { return OW_ABI_VERSION; }
const char* ow_last_error(void) { return g_err.c_str(); }
void ow_clear_error(void) { g_err.clear(); }

ow_pool* ow_pool_new(double sample_rate, size_t n_engines, int device, int preamp_kind) {
    try { return pool_create(sample_rate, n_engines, device, preamp_kind); }
    catch (const std::exception& ex) { set_err(std::string("ow_pool_new: ") + ex.what()); return nullptr; }
}
ow_pool* ow_pool_new_kinds(double sample_rate, size_t n_engines, int device, int preamp_kind, int power_amp_kind, int tremolo_kind) {
    try { return pool_create(sample_rate, n_engines, device, preamp_kind, power_amp_kind, tremolo_kind); }
    catch (const std::exception& ex) { set_err(std::string("ow_pool_new_kinds: ") + ex.what()); return nullptr; }
}
ow_pool* ow_pool_new_with(double sample_rate, size_t n_engines, int device, int preamp_kind, int power_amp_kind) {
    try { return pool_create(sample_rate, n_engines, device, preamp_kind, power_amp_kind); }
    catch (const std::exception& ex) { set_err(std::string("ow_pool_new_with: ") + ex.what()); return nullptr; }
}
void ow_pool_free(ow_pool* p) { pool_destroy(p); }
size_t ow_pool_size(const ow_pool* p) { return p ? p->I : 0; }
ow_engine* ow_pool_engine(ow_pool* p, size_t i) { return (p && i < p->I) ? p->engines[i] : nullptr; }
ow_pool* ow_engine_pool(ow_engine* e) { return e ? e->pool : nullptr; }
void* ow_pool_stream(ow_pool* p) { return p ? (void*)p->stream : nullptr; }
void ow_pool_set_profiling(ow_pool* p, int on) { if (p) p->profiling = on != 0; }
void ow_pool_last_kernel_ms(const ow_pool* p, float ms[5]) { for (int i = 0; i < 5; ++i) ms[i] = p ? p->last_ms[i] : 0.f; }

int ow_pool_set_sample_rate(ow_pool* p, double sr) {
    if (!p || !(sr > 0.0)) return -1;
    try {
        HIP_OK(hipSetDevice(p->device));
        upload_consts(p, sr, p->hc.preamp_kind);
        for (ow_engine* en : p->engines) {
            en->sr = sr;
            en->noise_on = false; en->thermal_gain = 1.0;   // set_sample_rate builds a new DkPreamp (engine.rs:276): noise off, gain 1.0
            en->rail_sag = true;                            // ... and a new PowerAmp (engine.rs:279): rail sag back on
            en->touch();
        }
        // voices keep their records (the reference keeps Voice objects too, engine.rs:272-286), chain objects are rebuilt
        std::vector<double> d0(p->I);
        for (size_t i = 0; i < p->I; ++i) d0[i] = p->engines[i]->depth.target;
        chain_init_range(p, 0, (int)p->I, INIT_RATE, d0);
        warm_up_range(p, 0, (int)p->I);
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_pool_set_sample_rate: ") + ex.what()); return -1; }
}

void ow_pool_reset(ow_pool* p) {
    if (!p) return;
    guarded("ow_pool_reset", [&] {
        HIP_OK(hipSetDevice(p->device));
        for (ow_engine* en : p->engines) engine_host_reset(en);
        chain_init_range(p, 0, (int)p->I, INIT_RESET, std::vector<double>(p->I, 0.0));
        warm_up_range(p, 0, (int)p->I);
    });
}

void ow_pool_ensure_buffer_capacity(ow_pool* p, size_t n) {
    if (!p || n <= p->Lcap) return;
    guarded("ow_pool_ensure_buffer_capacity", [&] {
        HIP_OK(hipSetDevice(p->device));
        HIP_OK(hipStreamSynchronize(p->stream));
        alloc_stream_buffers(p, n);
    });
}

void ow_pool_render(ow_pool* p, float* out_host, size_t out_stride, size_t len) {
    if (!p || len == 0) return;
    const bool ok = guarded("ow_pool_render", [&] {
        HIP_OK(hipSetDevice(p->device));
        if (p->inject_faults > 0) { --p->inject_faults; throw std::runtime_error("injected fault (ow_test_inject_render_faults)"); }
        if (len > p->Lcap) { HIP_OK(hipStreamSynchronize(p->stream)); alloc_stream_buffers(p, len); }  // auto-grow, engine.rs:430
        const bool hostprof = p->sw.host_profile;
        auto t0 = std::chrono::steady_clock::now();
        render_range(p, 0, (int)p->I, len, true, out_host, out_stride);
        steal_countdown_early(p, len);
        auto t1 = std::chrono::steady_clock::now();
        HIP_OK(hipStreamSynchronize(p->stream));
        auto t2 = std::chrono::steady_clock::now();
        post_render_host(p, 0, (int)p->I, len);
        auto t3 = std::chrono::steady_clock::now();
        collect_profile(p);
        if (p->d_vm && p->vm_bursts && p->vm_host_dirty && p->pipe_stream[1] && !p->vm_download_pending) {   // see ow_pool::ev_vm_up
            if (p->vm_upload_inflight) HIP_OK(hipStreamWaitEvent(p->pipe_stream[1], p->ev_vm_up, 0));
            __atomic_store_n(&p->vm_host_dirty, (uint8_t)0, __ATOMIC_RELAXED);
            HIP_OK(hipMemcpyAsync(p->d_vm, p->h_vm, sizeof(OwVm) * p->I, hipMemcpyHostToDevice, p->pipe_stream[1]));
            HIP_OK(hipEventRecord(p->ev_vm_up, p->pipe_stream[1]));
            p->vm_upload_inflight = true;
        }
        auto t4 = std::chrono::steady_clock::now();
        if (hostprof) {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            double* acc = p->hostprof_acc;
            acc[0] += ms(t0, t1); acc[1] += ms(t1, t2); acc[2] += ms(t2, t3); acc[3] += ms(t3, t4);
            if (ms(t0, t1) > 2.0 || ms(t2, t3) > 1.0)      // the blocks around a re-strike: where the host time between the kernels goes
                std::fprintf(stderr, "hostprof block %ld len %zu: launch %.3f wait %.3f post %.3f ms\n", p->hostprof_cnt, len, ms(t0, t1), ms(t1, t2), ms(t2, t3));
            if (++p->hostprof_cnt % 50 == 0) { std::fprintf(stderr, "hostprof I=%zu: launch %.3f wait %.3f post %.3f profile %.3f ms (mean of 50)\n", p->I, acc[0] / 50, acc[1] / 50, acc[2] / 50, acc[3] / 50); acc[0] = acc[1] = acc[2] = acc[3] = 0; }
        }
        p->last_len = len;
    });
    if (!ok) {
        // queued note events / setter targets of engines the failed render did not get to pack must survive: the next block scans again
        __atomic_store_n(&p->dirty_any, (uint8_t)1, __ATOMIC_RELAXED);
        p->args_stale = true; p->lists_valid = false; p->steal_counted = false;
        // "never fails, degrades to silence" (engine.rs:450-458 does the same for numeric failure): every row of the
        // caller's block is written.  Drain the stream first so that an output copy already queued cannot land after the zeros.
        for (int k = 0; k < OW_MAX_STAGES; ++k) if (p->pipe_stream[k]) hipStreamSynchronize(p->pipe_stream[k]);
        if (out_host && out_stride >= len)
            for (size_t e = 0; e < p->I; ++e) std::memset(out_host + e * out_stride, 0, len * sizeof(float));
        if (p->d_out && len <= p->Lcap) { hipMemset(p->d_out, 0, sizeof(float) * len * p->I); p->out_ld = len; }   // the HBM copy of the block too
    }
}

const float* ow_pool_device_output(const ow_pool* p, size_t* stride) {
    if (!p) return nullptr;
    if (stride) *stride = p->out_ld;
    return p->d_out;
}

// Shared tremolo trajectory of (device, the chain rate of host rate `sample_rate`): make its first `seconds` exist now (blocking), e.g.
// when a host instantiates the plugin, so that no engine ever waits for the single oscillator that extends it.  Returns the number of
// samples the store holds afterwards, <0 on error.
long long ow_tremolo_prefetch(double sample_rate, int device, double seconds) {
    try {
        if (!(sample_rate > 0.0) || !(seconds >= 0.0)) throw std::runtime_error("bad argument");
        HIP_OK(hipSetDevice(device));
        std::unique_ptr<OwConsts> hc(new OwConsts()), k48(new OwConsts());
        owhip::build_consts(*hc, sample_rate, OW_PREAMP_LEGACY8);
        owhip::build_consts(*k48, 24000.0, OW_PREAMP_LEGACY8);
        const Switches sw = Switches::from_env();
        std::shared_ptr<TremTraj> t = traj_acquire(device, *hc, *k48, sw.trem_cache);
        const size_t want = (size_t)std::min(seconds * hc->os_sr, (double)t->cap_max);
        t->grow_to(want);                                    // instantiation time: the place to allocate
        hipEvent_t ev;
        { std::lock_guard<std::mutex> lk(t->mu); ev = t->cover(want, 0); }
        if (ev) HIP_OK(hipEventSynchronize(ev));
        std::lock_guard<std::mutex> lk(t->mu);
        return (long long)std::max(t->done, std::min(want, t->len));     // (the store may hold more: the background lead)
    } catch (const std::exception& ex) { set_err(std::string("ow_tremolo_prefetch: ") + ex.what()); return -1; }
}

// ---- trajectory persistence (ow_tremolo_export / ow_tremolo_import) -----------------------------------------------------------------
// The reference's expensive start-up states are in-process caches (OnceLock: dk_preamp/melange_adapter.rs:12-29; Tremolo::new settles in
// its constructor, tremolo.rs:92-102), paid once per process.  Here the analogue of that cost is the trajectory itself: a fresh process
// that renders offline faster than the one oscillator steps waits for it.  A host may therefore keep the store across processes: export
// writes r_ldr[0 .. L) (L = the store's length cut to a checkpoint boundary), the checkpoints, the fallback list and the settled rows;
// import loads them into the store of (device, chain rate) -- after checking that THIS library would have produced them: same build, same
// tremolo constants, payload checksum, and the first and the last 4 096-sample segment regenerated on the device by the product kernel
// from the file's own settled rows / checkpoint and compared bit for bit (samples and the checkpoint behind them).
namespace {
struct TrajFileHeader {
    char magic[8];                  // "OWTRAJ1\0"
    uint32_t abi, ck;               // OW_ABI_VERSION, OW_TRAJ_CK
    uint64_t build_id, consts_hash, rate_bits, len, be_settle, payload_sum;
    double settled[18];
    uint64_t reserved[4];
};
uint64_t fnv64(const void* data, size_t bytes, uint64_t h = 0xCBF29CE484222325ull) {
    const unsigned char* b = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 0x100000001B3ull;
    return h;
}
uint64_t words_sum(const void* data, size_t bytes, uint64_t h) {     // order-dependent mix over 64-bit words (payloads are arrays of 8-byte items)
    const uint64_t* w = static_cast<const uint64_t*>(data);
    for (size_t i = 0; i < bytes / 8; ++i) { h ^= w[i]; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
    return h;
}
uint64_t traj_build_id() { static const char stamp[] = "openwurli-hip " __DATE__ " " __TIME__; return fnv64(stamp, sizeof stamp) ^ (uint64_t)OW_ABI_VERSION; }
uint64_t traj_consts_hash(const OwConsts& c) {
    uint64_t h = fnv64(c.t_a_neg, sizeof c.t_a_neg);
    h = fnv64(c.t_s, sizeof c.t_s, h); h = fnv64(c.t_k, sizeof c.t_k, h); h = fnv64(c.t_s_ni, sizeof c.t_s_ni, h);
    h = fnv64(c.t_a_neg_be, sizeof c.t_a_neg_be, h); h = fnv64(c.t_s_be, sizeof c.t_s_be, h); h = fnv64(c.t_k_be, sizeof c.t_k_be, h);
    h = fnv64(c.t_s_ni_be, sizeof c.t_s_ni_be, h);
    const double tail[4] = {c.ldr_attack, c.ldr_release, c.ln_r_max, c.ln_min_minus_max};
    return fnv64(tail, sizeof tail, h);
}
struct FileCloser { std::FILE* f = nullptr; ~FileCloser() { if (f) std::fclose(f); } };
// oscillator rows (I = 1 layout) in front of sample k * OW_TRAJ_CK, from checkpoint k (v[7] ip[4] ipp[4] env) and the sample before it
void traj_state_from_ckpt(const double* ck, double r_before, double* rows) {
    for (int i = 0; i < 7; ++i) rows[CS_T_V + i] = ck[i];
    for (int i = 0; i < 4; ++i) { rows[CS_T_I + i] = ck[7 + i]; rows[CS_T_IP + i] = ck[11 + i]; }
    rows[CS_T_ENV] = ck[15]; rows[CS_T_RLDR] = r_before;
    const uint64_t z = 0; std::memcpy(&rows[CS_T_BE], &z, 8);
}
}  // namespace

long long ow_tremolo_export(double sample_rate, int device, const char* path) {
    try {
        if (!(sample_rate > 0.0) || !path) throw std::runtime_error("bad argument");
        HIP_OK(hipSetDevice(device));
        std::unique_ptr<OwConsts> hc(new OwConsts()), k48(new OwConsts());
        owhip::build_consts(*hc, sample_rate, OW_PREAMP_LEGACY8);
        owhip::build_consts(*k48, 24000.0, OW_PREAMP_LEGACY8);
        const Switches sw = Switches::from_env();
        std::shared_ptr<TremTraj> t = traj_acquire(device, *hc, *k48, sw.trem_cache);
        TrajFileHeader h;
        std::memset(&h, 0, sizeof h);
        std::memcpy(h.magic, "OWTRAJ1", 8);
        h.abi = OW_ABI_VERSION; h.ck = OW_TRAJ_CK; h.build_id = traj_build_id(); h.consts_hash = traj_consts_hash(*hc);
        std::memcpy(&h.rate_bits, &hc->os_sr, 8);
        std::vector<double> r, ck;
        std::vector<unsigned long long> be(1 + OW_TRAJ_BE_CAP);
        {
            std::lock_guard<std::mutex> lk(t->mu);
            HIP_OK(hipStreamSynchronize(t->stream));                   // everything enqueued has been produced
            const size_t L = t->len / OW_TRAJ_CK * OW_TRAJ_CK;
            if (L == 0) throw std::runtime_error("the store holds less than one checkpoint segment");
            h.len = L; h.be_settle = t->be_settle;
            r.resize(L); ck.resize((L / OW_TRAJ_CK + 1) * OW_TRAJ_CKD);
            HIP_OK(hipMemcpy(r.data(), t->d_r, sizeof(double) * L, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(ck.data(), t->d_ckpt, sizeof(double) * ck.size(), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(be.data(), t->d_be, sizeof(unsigned long long) * be.size(), hipMemcpyDeviceToHost));
        }
        // events of samples beyond L belong to what is not exported
        { size_t k = 0; const size_t n = (size_t)std::min<unsigned long long>(be[0], OW_TRAJ_BE_CAP);
          for (size_t i = 0; i < n; ++i) if (be[1 + i] < h.len) be[1 + k++] = be[1 + i];
          for (size_t i = k; i < OW_TRAJ_BE_CAP; ++i) be[1 + i] = ~0ull;
          be[0] = k; }
        traj_state_from_ckpt(ck.data(), 1000000.0, h.settled);         // the settled rows ARE checkpoint 0 (the cell at rest: r_ldr = 1 MOhm)
        std::memcpy(&h.settled[17], &h.be_settle, 8);
        h.payload_sum = words_sum(be.data(), sizeof(unsigned long long) * be.size(), words_sum(ck.data(), sizeof(double) * ck.size(), words_sum(r.data(), sizeof(double) * r.size(), 0x0123456789ABCDEFull)));
        FileCloser fc;
        fc.f = std::fopen(path, "wb");
        if (!fc.f) throw std::runtime_error(std::string("cannot open ") + path);
        if (std::fwrite(&h, sizeof h, 1, fc.f) != 1 || std::fwrite(r.data(), sizeof(double), r.size(), fc.f) != r.size() ||
            std::fwrite(ck.data(), sizeof(double), ck.size(), fc.f) != ck.size() || std::fwrite(be.data(), sizeof(unsigned long long), be.size(), fc.f) != be.size())
            throw std::runtime_error("short write");
        return (long long)h.len;
    } catch (const std::exception& ex) { (void)hipGetLastError(); set_err(std::string("ow_tremolo_export: ") + ex.what()); return -1; }
}

long long ow_tremolo_import(double sample_rate, int device, const char* path) {
    try {
        if (!(sample_rate > 0.0) || !path) throw std::runtime_error("bad argument");
        HIP_OK(hipSetDevice(device));
        std::unique_ptr<OwConsts> hc(new OwConsts()), k48(new OwConsts());
        owhip::build_consts(*hc, sample_rate, OW_PREAMP_LEGACY8);
        owhip::build_consts(*k48, 24000.0, OW_PREAMP_LEGACY8);
        FileCloser fc;
        fc.f = std::fopen(path, "rb");
        if (!fc.f) throw std::runtime_error(std::string("cannot open ") + path);
        TrajFileHeader h;
        if (std::fread(&h, sizeof h, 1, fc.f) != 1 || std::memcmp(h.magic, "OWTRAJ1", 8) != 0) throw std::runtime_error("not a trajectory file");
        uint64_t rate_bits; std::memcpy(&rate_bits, &hc->os_sr, 8);
        if (h.abi != OW_ABI_VERSION || h.ck != OW_TRAJ_CK || h.build_id != traj_build_id()) throw std::runtime_error("written by another build of the library");
        if (h.rate_bits != rate_bits) throw std::runtime_error("written for another chain rate");
        if (h.consts_hash != traj_consts_hash(*hc)) throw std::runtime_error("written with other tremolo constants");
        const size_t L = (size_t)h.len;
        if (L == 0 || L % OW_TRAJ_CK != 0 || L > (size_t)4.0e9) throw std::runtime_error("bad length");
        std::vector<double> r(L), ck((L / OW_TRAJ_CK + 1) * OW_TRAJ_CKD);
        std::vector<unsigned long long> be(1 + OW_TRAJ_BE_CAP);
        if (std::fread(r.data(), sizeof(double), r.size(), fc.f) != r.size() || std::fread(ck.data(), sizeof(double), ck.size(), fc.f) != ck.size() ||
            std::fread(be.data(), sizeof(unsigned long long), be.size(), fc.f) != be.size())
            throw std::runtime_error("truncated file");
        if (h.payload_sum != words_sum(be.data(), sizeof(unsigned long long) * be.size(), words_sum(ck.data(), sizeof(double) * ck.size(), words_sum(r.data(), sizeof(double) * r.size(), 0x0123456789ABCDEFull))))
            throw std::runtime_error("checksum mismatch");
        // ---- would this library have produced it?  Segment 0 from the file's settled rows, the last segment from its checkpoint.
        {
            DevMem dk, dstate, dr, dck, dbe;
            dk.alloc(sizeof(OwConsts)); dstate.alloc(sizeof(double) * 18); dr.alloc(sizeof(double) * (OW_TRAJ_CK + 64));
            const size_t nck = TremTraj::ckpt_doubles(L);
            dck.alloc(sizeof(double) * nck); dbe.alloc(sizeof(unsigned long long) * (1 + OW_TRAJ_BE_CAP));
            HIP_OK(hipMemcpy(dk.p, hc.get(), sizeof(OwConsts), hipMemcpyHostToDevice));
            std::vector<double> got(OW_TRAJ_CK), gck(OW_TRAJ_CKD);
            const size_t K = L / OW_TRAJ_CK;
            for (int pass = 0; pass < 2; ++pass) {
                const size_t seg = pass == 0 ? 0 : K - 1;
                if (pass == 1 && seg == 0) break;
                double rows[18];
                if (seg == 0) { std::memcpy(rows, h.settled, sizeof rows); const uint64_t z = 0; std::memcpy(&rows[17], &z, 8); }
                else traj_state_from_ckpt(ck.data() + seg * OW_TRAJ_CKD, r[seg * OW_TRAJ_CK - 1], rows);
                HIP_OK(hipMemcpy(dstate.p, rows, sizeof rows, hipMemcpyHostToDevice));
                HIP_OK(hipMemset(dbe.p, 0, sizeof(unsigned long long) * (1 + OW_TRAJ_BE_CAP)));
                owdev::k_trem_traj_extend_row<<<dim3(1), dim3(64)>>>(dk.as<OwConsts>(), dstate.as<double>(), dr.as<double>(), (long long)(seg * OW_TRAJ_CK),
                                                                      (long long)OW_TRAJ_CK, dck.as<double>(), dbe.as<unsigned long long>());
                HIP_OK(hipGetLastError());
                HIP_OK(hipMemcpy(got.data(), dr.p, sizeof(double) * OW_TRAJ_CK, hipMemcpyDeviceToHost));
                HIP_OK(hipMemcpy(gck.data(), dck.as<double>() + (seg + 1) * OW_TRAJ_CKD, sizeof(double) * OW_TRAJ_CKD, hipMemcpyDeviceToHost));
                if (std::memcmp(got.data(), r.data() + seg * OW_TRAJ_CK, sizeof(double) * OW_TRAJ_CK) != 0 ||
                    std::memcmp(gck.data(), ck.data() + (seg + 1) * OW_TRAJ_CKD, sizeof(double) * OW_TRAJ_CKD) != 0)
                    throw std::runtime_error("the regenerated segment differs from the file's");
            }
        }
        // ---- the settled rows spare the new store its settle; then the store takes what it does not have yet
        const Switches sw = Switches::from_env();
        if (sw.trem_cache) {
            TremSettled ts;
            std::memcpy(ts.rows, h.settled, sizeof ts.rows);
            std::lock_guard<std::mutex> lk(g_mel_mu);
            g_trem_settled.emplace(std::make_pair(device, rate_bits), ts);        // (an existing entry stays: it was computed here)
        }
        std::shared_ptr<TremTraj> t = traj_acquire(device, *hc, *k48, sw.trem_cache);
        if (t->be_settle != h.be_settle) throw std::runtime_error("settled state differs from this library's");
        if (L > t->cap_max) throw std::runtime_error("longer than the store's configured capacity (ow_tremolo_configure)");
        t->grow_to(L);
        std::lock_guard<std::mutex> lk(t->mu);
        HIP_OK(hipStreamSynchronize(t->stream));
        if (L > t->cap) throw std::runtime_error("the store could not grow to the file's length");
        if (L <= t->len) return 0;                                         // the store already holds more
        const size_t from = t->len / OW_TRAJ_CK * OW_TRAJ_CK;               // whole segments from the checkpoint at or below the store's end (same bits where they overlap)
        HIP_OK(hipMemcpy(t->d_r + from, r.data() + from, sizeof(double) * (L - from), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(t->d_ckpt + from / OW_TRAJ_CK * OW_TRAJ_CKD, ck.data() + from / OW_TRAJ_CK * OW_TRAJ_CKD, sizeof(double) * (L / OW_TRAJ_CK - from / OW_TRAJ_CK + 1) * OW_TRAJ_CKD, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(t->d_be, be.data(), sizeof(unsigned long long) * be.size(), hipMemcpyHostToDevice));
        double rows[18];
        traj_state_from_ckpt(ck.data() + L / OW_TRAJ_CK * OW_TRAJ_CKD, r[L - 1], rows);
        HIP_OK(hipMemcpy(t->d_state, rows, sizeof rows, hipMemcpyHostToDevice));
        t->len = L; t->done = L;
        t->target = std::max(t->target, L);
        return (long long)L;
    } catch (const std::exception& ex) { (void)hipGetLastError(); set_err(std::string("ow_tremolo_import: ") + ex.what()); return -1; }
}

// Capacity and lead of the trajectory stores of `device` (seconds of audio): capacity_seconds = how old an engine may grow (time since
// new / reset / set_sample_rate) before it leaves the shared trajectory for an oscillator of its own (default 1 800; <= 0 restores it);
// lead_seconds = how far the store is kept ahead of its oldest reader in the background (default 60; 0 = only the block ahead; < 0
// restores the default).  The buffers are NOT reserved at that size: a store starts at 150 s (115 MB at 96 kHz; all of it for pools of
// >= 4 096 engines) and doubles on a helper thread well before a reader gets there.  Applies to stores created afterwards (a store is
// created by the first engine of its device and chain rate) and raises / lowers the limits of the existing ones.  0 on success.
int ow_tremolo_configure(int device, double capacity_seconds, double lead_seconds) {
    try {
        std::vector<std::shared_ptr<TremTraj>> live;
        {
            std::lock_guard<std::mutex> lk(g_traj_mu);
            TrajConfig& c = g_traj_cfg[device];
            c.seconds = capacity_seconds > 0.0 ? capacity_seconds : 0.0;
            c.lead = lead_seconds >= 0.0 ? lead_seconds : -1.0;
            for (auto& kv : traj_registry()) if (kv.first.first == device) live.push_back(kv.second);
        }
        for (auto& t : live) {
            std::lock_guard<std::mutex> lk(t->mu);
            const double secs = capacity_seconds > 0.0 ? capacity_seconds : OW_TRAJ_DEFAULT_SECONDS, lead = lead_seconds >= 0.0 ? lead_seconds : OW_TRAJ_DEFAULT_LEAD;
            const size_t want = ((size_t)std::min(secs * t->os_sr, 4.0e9) + OW_TRAJ_CK - 1) / OW_TRAJ_CK * OW_TRAJ_CK;
            t->cap_max = std::max(want, t->cap);             // never below what is already allocated (engines may stand there)
            t->lead = (size_t)std::min(lead * t->os_sr, (double)t->cap_max);
        }
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_tremolo_configure: ") + ex.what()); return -1; }
}

int ow_pool_read_voice_sum(ow_pool* p, double* out_host, size_t out_stride, size_t len) {
    if (!p || !out_host || len > p->Lcap) return -1;
    try {
        HIP_OK(hipSetDevice(p->device));
        const size_t I = p->I;
        std::vector<double> a(I * len), b(I * len);
        HIP_OK(hipMemcpy2D(a.data(), len * sizeof(double), p->d_sum, p->Lcap * sizeof(double), len * sizeof(double), I, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy2D(b.data(), len * sizeof(double), p->d_sum + I * p->Lcap, p->Lcap * sizeof(double), len * sizeof(double), I, hipMemcpyDeviceToHost));
        for (size_t e = 0; e < I; ++e) {
            const OwEngineArgs& ar = p->h_args[e];
            for (size_t n = 0; n < len; ++n) {
                double x = 0.0;
                if (!p->h_eout[e].sum_nonfinite) {
                    if (ar.main_mask) x = a[e * len + n];
                    if (ar.steal_mask) x += b[e * len + n];
                }
                out_host[e * out_stride + n] = x;
            }
        }
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_pool_read_voice_sum: ") + ex.what()); return -1; }
}

int ow_pool_read_preamp_out(ow_pool* p, double* out_host, size_t out_stride, size_t n_os) {
    if (!p || !out_host || n_os > 2 * p->Lcap) return -1;
    try {
        HIP_OK(hipSetDevice(p->device));
        const size_t I = p->I;
        std::vector<double> a(I * n_os);
        HIP_OK(hipMemcpy(a.data(), p->d_pre, sizeof(double) * I * n_os, hipMemcpyDeviceToHost));  // [n_os][I]
        for (size_t e = 0; e < I; ++e)
            for (size_t n = 0; n < n_os; ++n) out_host[e * out_stride + n] = a[n * I + e];
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_pool_read_preamp_out: ") + ex.what()); return -1; }
}

int ow_pool_read_tremolo_r(ow_pool* p, double* out_host, size_t out_stride, size_t n_os) {
    if (!p || !out_host || n_os > 2 * p->Lcap) return -1;
    try {
        HIP_OK(hipSetDevice(p->device));
        const size_t I = p->I;
        std::vector<double> a(I * n_os);
        const double* src = p->d_rbuf + (size_t)p->rb_cur * (2 * p->Lcap * I);   // the half the last block consumed, [n_os][I]
        HIP_OK(hipMemcpy(a.data(), src, sizeof(double) * I * n_os, hipMemcpyDeviceToHost));
        std::vector<double> tr;
        if (p->traj && p->n_on_traj) {     // engines on the shared trajectory: the n_os samples below their present t
            if ((long long)n_os > p->last_n_os) throw std::runtime_error("more samples than the last block consumed");
            // (the samples the last block consumed are complete -- its kernels waited for them; no wait for the store's stream, which
            // the background extension keeps busy)
            HIP_OK(hipStreamSynchronize(p->stream));
            DevMem g;
            g.alloc(sizeof(double) * I * n_os);
            const double* base;
            { std::lock_guard<std::mutex> lk(p->traj->mu); base = p->traj->d_r; }
            // the block consumed [t_end - last_n_os, t_end); its first n_os samples are asked for
            owdev::k_trem_traj_gather<<<dim3((unsigned)((I * n_os + 255) / 256)), dim3(256), 0, p->stream>>>(
                base + p->trem_clock - (p->last_n_os - (long long)n_os), p->d_birth, (int)I, (long long)n_os, g.as<double>());
            HIP_OK(hipGetLastError());
            tr.resize(I * n_os);
            HIP_OK(hipMemcpyAsync(tr.data(), g.p, sizeof(double) * I * n_os, hipMemcpyDeviceToHost, p->stream));
            HIP_OK(hipStreamSynchronize(p->stream));
        }
        for (size_t e = 0; e < I; ++e) {
            if (!tr.empty() && p->h_birth[e] != OW_OFF_TRAJ) { std::memcpy(out_host + e * out_stride, tr.data() + e * n_os, sizeof(double) * n_os); continue; }
            for (size_t n = 0; n < n_os; ++n) out_host[e * out_stride + n] = a[n * I + p->h_lead[e]];   // the column of the engine's phase group
        }
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_pool_read_tremolo_r: ") + ex.what()); return -1; }
}

// ---- engines ------------------------------------------------------------------------------------
ow_engine* ow_engine_new(double sample_rate, int device, int preamp_kind) {
    ow_pool* p = ow_pool_new(sample_rate, 1, device, preamp_kind);
    if (!p) return nullptr;
    p->engines[0]->owns_pool = true;
    return p->engines[0];
}
ow_engine* ow_engine_new_kinds(double sample_rate, int device, int preamp_kind, int power_amp_kind, int tremolo_kind) {
    ow_pool* p = ow_pool_new_kinds(sample_rate, 1, device, preamp_kind, power_amp_kind, tremolo_kind);
    if (!p) return nullptr;
    p->engines[0]->owns_pool = true;
    return p->engines[0];
}
ow_engine* ow_engine_new_with(double sample_rate, int device, int preamp_kind, int power_amp_kind) {
    ow_pool* p = ow_pool_new_with(sample_rate, 1, device, preamp_kind, power_amp_kind);
    if (!p) return nullptr;
    p->engines[0]->owns_pool = true;
    return p->engines[0];
}
void ow_engine_free(ow_engine* e) { if (e && e->owns_pool) pool_destroy(e->pool); }
// engine.rs:406-420.  No-ops / zeros on the behavioural amp, which has no separable rails (power_amp.rs:262-272).
void ow_engine_set_rail_sag(ow_engine* e, int on) {
    if (!e || !e->pool || e->pool->power_amp_kind != OW_POWER_AMP_MELANGE) return;
    if (e->rail_sag != (on != 0)) { e->rail_sag = on != 0; e->touch(); }
}
int ow_engine_rail_sag_enabled(const ow_engine* e) { return (e && e->pool && e->pool->power_amp_kind == OW_POWER_AMP_MELANGE && e->rail_sag) ? 1 : 0; }
void ow_engine_power_amp_diag(const ow_engine* e, ow_power_amp_diag* d) {
    if (!d) return;
    std::memset(d, 0, sizeof *d);
    d->rail_pos_volts = 22.5; d->rail_neg_volts = 22.5;
    if (!e || !e->pool || e->pool->power_amp_kind != OW_POWER_AMP_MELANGE) return;
    ow_pool* p = e->pool;
    double col[owdev::PAS_COUNT];
    if (hipSetDevice(p->device) != hipSuccess) return;
    hipStreamSynchronize(p->stream);
    if (hipMemcpy2D(col, sizeof(double), p->d_pa + e->index, sizeof(double) * p->I, sizeof(double), owdev::PAS_COUNT, hipMemcpyDeviceToHost) != hipSuccess) return;
    auto u64 = [&](int r) { uint64_t b; std::memcpy(&b, &col[r], 8); return b; };
    d->clamp_count = u64(owdev::PAS_CLAMP); d->nr_max_iter_count = u64(owdev::PAS_NRMAX); d->peak_output_volts = col[owdev::PAS_PEAK];
    d->nan_resets = u64(owdev::PAS_NAN); d->guard_resets = u64(owdev::PAS_GUARD);
    if (e->rail_sag) { d->rail_pos_volts = col[owdev::PAS_RAILP]; d->rail_neg_volts = col[owdev::PAS_RAILN]; }
}

void ow_engine_set_sample_rate(ow_engine* e, double sr) {
    if (!e) return;
    // engines of a pool share one rate (lane = engine kernels read one constant block): the whole pool follows
    ow_pool_set_sample_rate(e->pool, sr);
}

void ow_engine_reset(ow_engine* e) {
    if (!e) return;
    guarded("ow_engine_reset", [&] {
        ow_pool* p = e->pool;
        HIP_OK(hipSetDevice(p->device));
        engine_host_reset(e);
        chain_init_range(p, (int)e->index, 1, INIT_RESET, std::vector<double>(1, 0.0));
        warm_up_range(p, (int)e->index, 1);
    });
}

void ow_engine_warm_up(ow_engine* e) {
    if (!e) return;
    guarded("ow_engine_warm_up", [&] { HIP_OK(hipSetDevice(e->pool->device)); warm_up_range(e->pool, (int)e->index, 1); });
}

void ow_engine_ensure_buffer_capacity(ow_engine* e, size_t n) { if (e) ow_pool_ensure_buffer_capacity(e->pool, n); }

// (the state machine itself: ow_vm.h -- the same functions run on the device for bursts, k_vm_events)
void ow_engine_note_on(ow_engine* e, uint8_t note_in, float velocity) {  // engine.rs:299-338
    if (!e) return;
    vm_host_current(e); vm_host_changed(e);
    vm_note_on(*e->vm, *e, note_in, velocity, owhip::sat_u32(e->sr * 0.005));
    e->mark();
}
void ow_engine_note_off(ow_engine* e, uint8_t note_in) {  // engine.rs:340-359
    if (!e) return;
    vm_host_current(e); vm_host_changed(e);
    vm_note_off(*e->vm, *e, note_in);
}
void ow_engine_set_sustain(ow_engine* e, int held) {  // engine.rs:361-374
    if (!e) return;
    vm_host_current(e); vm_host_changed(e);
    vm_set_sustain(*e->vm, *e, held != 0);
}

void ow_engine_set_volume(ow_engine* e, double v) { if (e) { e->volume.set_target(v); if (e->volume.pending) e->touch(); } }
void ow_engine_set_tremolo_depth(ow_engine* e, double d) { if (e) { e->depth.set_target(d); if (e->depth.pending) e->touch(); } }
void ow_engine_set_speaker_character(ow_engine* e, double c) { if (e) { e->spk.set_target(c); if (e->spk.pending) e->touch(); } }
void ow_engine_set_mlp_enabled(ow_engine* e, int on) { if (e) { vm_host_current(e); e->vm->mlp_enabled = on != 0 ? 1 : 0; vm_host_changed(e); } }
// Thermal noise of the melange preamp's main state; no-ops on the legacy solver (dk_preamp_legacy.rs:262-265)
void ow_engine_set_noise_enabled(ow_engine* e, int on) {
    if (!e || !e->pool || e->pool->hc.preamp_kind != OW_PREAMP_MELANGE12) return;
    if (e->noise_on != (on != 0)) { e->noise_on = on != 0; e->touch(); }
}
void ow_engine_set_noise_gain(ow_engine* e, double gain) {
    if (!e || !e->pool || e->pool->hc.preamp_kind != OW_PREAMP_MELANGE12) return;
    if (e->thermal_gain != gain) { e->thermal_gain = gain; e->touch(); }
}
void ow_engine_set_noise_seed(ow_engine* e, uint64_t seed) {
    if (!e || !e->pool || !e->pool->d_noise) return;
    guarded("ow_engine_set_noise_seed", [&] {
        ow_pool* p = e->pool;
        HIP_OK(hipSetDevice(p->device));
        const uint64_t resolved = seed ? seed : process_noise_seed();
        HIP_OK(hipMemcpyAsync(p->d_noise + (size_t)NZ_SEED * p->I + e->index, &resolved, sizeof resolved, hipMemcpyHostToDevice, p->stream));
        owdev::k_mel_noise_seed<<<dim3(1), dim3(64), 0, p->stream>>>(p->d_noise, (int)p->I, (int)e->index, 1);
        HIP_OK(hipStreamSynchronize(p->stream));
    });
}

void ow_engine_render(ow_engine* e, float* out, size_t len) {
    if (!e || !out || len == 0) return;
    if (e->pool->I != 1) { set_err("ow_engine_render: engine belongs to a multi-engine pool; use ow_pool_render"); std::memset(out, 0, len * sizeof(float)); return; }
    ow_pool_render(e->pool, out, len, len);      // writes silence itself when the render fails
}

void ow_engine_get_diag(const ow_engine* e, ow_diag* d) {
    if (!e || !d) return;
    std::memset(d, 0, sizeof *d);
    vm_host_current(e);
    const OwVm& v = *e->vm;
    d->active_voices = (uint32_t)__builtin_popcountll(~v.st_mask[OW_VOICE_FREE]);
    d->held_voices = (uint32_t)__builtin_popcountll(v.st_mask[OW_VOICE_HELD]);
    d->sustained_voices = (uint32_t)__builtin_popcountll(v.st_mask[OW_VOICE_SUSTAINED]);
    d->releasing_voices = (uint32_t)__builtin_popcountll(v.st_mask[OW_VOICE_RELEASING]);
    d->steal_voices = (uint32_t)__builtin_popcountll(v.has_steal);
    d->sustain_held = v.sustain_held ? 1 : 0;
    d->nan_guard_fires = e->nan_guard_fires;
    d->output_nan_resets = e->output_nan_resets;
    ow_pool* p = e->pool;
    double diag = 0.0;
    if (p && hipSetDevice(p->device) == hipSuccess &&
        hipMemcpy(&diag, p->d_cs + (size_t)CS_DIAG * p->I + e->index, sizeof(double), hipMemcpyDeviceToHost) == hipSuccess) {
        uint64_t bits;
        std::memcpy(&bits, &diag, 8);
        d->preamp_nan_resets = (uint32_t)(bits >> 32);
        double be = 0.0;   // counts the block-ahead samples too
        if (p->traj && p->h_birth[e->index] != OW_OFF_TRAJ) {
            d->tremolo_be_fallbacks = p->traj->be_count_at(p->trem_clock - p->h_birth[e->index]);
        } else if (hipMemcpy(&be, p->d_cs + (size_t)CS_T_BE * p->I + p->h_lead[e->index], sizeof(double), hipMemcpyDeviceToHost) == hipSuccess) {
            std::memcpy(&bits, &be, 8);
            d->tremolo_be_fallbacks = bits;
        }
    }
}
int ow_engine_slot_state(const ow_engine* e, int slot) { vm_host_current(e); return (e && slot >= 0 && slot < OW_MAX_VOICES) ? e->state_of(slot) : -1; }
int ow_engine_slot_note(const ow_engine* e, int slot) { vm_host_current(e); return (e && slot >= 0 && slot < OW_MAX_VOICES) ? e->vm->midi_of[slot] : -1; }
int ow_engine_has_steal_voice_for(const ow_engine* e, uint8_t note) {
    if (!e) return 0;
    vm_host_current(e);
    for (int i = 0; i < OW_MAX_VOICES; ++i) if (e->vm->midi_of[i] == note && ((e->vm->has_steal >> i) & 1ull)) return 1;
    return 0;
}

static void midi_apply_one(ow_pool* p, const ow_midi_event& ev) {
    ow_engine* e = p->engines[ev.engine];
    switch (ev.type) {
        case 0: ow_engine_note_on(e, ev.note, ev.value); break;
        case 1: ow_engine_note_off(e, ev.note); break;
        case 2: ow_engine_set_sustain(e, ev.value >= 0.5f); break;
        default: break;
    }
}

// A burst on the device (ow_vm.h / ow_vm_kernels.h).  The list must be grouped by engine (the usual layout of a batched script; verified
// by the caller) and the ops it queues must fit the engines' fixed queues; false = not taken (the host path does it).
static bool midi_burst_on_device(ow_pool* p, const ow_midi_event* ev, size_t n) {
    const size_t I = p->I;
    HIP_OK(hipSetDevice(p->device));
    hipStream_t st = p->stream;
    if (!p->d_vm) {                                                // first burst: the device side of the state machine (instantiation-class work)
        HIP_OK(hipMalloc(&p->d_vm, sizeof(OwVm) * I));
        HIP_OK(hipMalloc(&p->d_ops_fix, sizeof(OwOp) * OW_VM_OPS_MAX * I));
        HIP_OK(hipMalloc(&p->d_ev_begin, sizeof(uint32_t) * 2 * I));
        HIP_OK(hipMalloc(&p->d_vm_ovf, sizeof(uint32_t)));
        HIP_OK(hipHostMalloc(&p->h_vm_ovf, sizeof(uint32_t)));
        p->vm_host_dirty = 1;
    }
    if (__atomic_load_n(&p->host_ops_any, __ATOMIC_RELAXED)) return false;   // ops queued on the host come first in their engines' queues: host path
    // the events: straight from the caller's block when it is pinned (ow_host_alloc), else through a pinned staging copy made by the workers
    const ow_midi_event* src = (const ow_midi_event*)host_block_device_ptr(ev, sizeof(ow_midi_event) * n) ? ev : nullptr;
    if (n > p->ev_cap) {
        if (p->d_ev) hipFree(p->d_ev);
        if (p->h_ev) hipHostFree(p->h_ev);
        p->d_ev = nullptr; p->h_ev = nullptr; p->ev_cap = 0;
        const size_t cap = std::max<size_t>(n, (size_t)2 * OW_MAX_VOICES * I);      // a whole-keyboard re-strike of every engine
        HIP_OK(hipMalloc(&p->d_ev, sizeof(ow_midi_event) * cap));
        HIP_OK(hipHostMalloc(&p->h_ev, sizeof(ow_midi_event) * cap));
        p->ev_cap = cap;
    }
    if (!src) {
        const size_t T = std::min<size_t>(effective_cpus(), OW_MAX_SLICES);
        auto copy = [&](size_t t) { const size_t a = n * t / T, b = n * (t + 1) / T; std::memcpy(p->h_ev + a, ev + a, sizeof(ow_midi_event) * (b - a)); };
        Workers::get().each(T, copy);
        src = p->h_ev;
    }
    vm_wait_download(p);
    const uint32_t e_lo = std::min<uint32_t>(ev[0].engine, (uint32_t)I), e_hi = std::min<uint32_t>(ev[n - 1].engine + 1u, (uint32_t)I);
    if (e_lo >= e_hi) return false;                                // nothing addressed to this pool -- if the list is grouped, which only the host path checks here
    if (p->vm_upload_inflight) { HIP_OK(hipStreamWaitEvent(st, p->ev_vm_up, 0)); p->vm_upload_inflight = false; }
    if (p->vm_host_dirty) {
        HIP_OK(hipMemcpyAsync(p->d_vm, p->h_vm, sizeof(OwVm) * I, hipMemcpyHostToDevice, st));
        p->vm_host_dirty = 0;
    }
    HIP_OK(hipMemcpyAsync(p->d_ev, src, sizeof(ow_midi_event) * n, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(p->d_ev_begin, 0, sizeof(uint32_t) * 2 * I, st));
    HIP_OK(hipMemsetAsync(p->d_vm_ovf, 0, sizeof(uint32_t), st));
    owdev::k_vm_index<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(p->d_ev, n, p->d_ev_begin, p->d_ev_begin + I, (uint32_t)I, p->d_vm_ovf);
    const uint32_t fade = owhip::sat_u32(p->engines[0]->sr * 0.005);
    owdev::k_vm_events<<<dim3((e_hi - e_lo + 63) / 64), dim3(64), 0, st>>>(p->d_vm, p->d_ev, p->d_ev_begin, p->d_ev_begin + I, p->d_ops_fix, e_lo, e_hi, fade, p->d_vm_ovf);
    HIP_OK(hipGetLastError());
    // a queue that overflowed (more than OW_VM_OPS_MAX slot ops for one engine between two renders) lost ops, a list that is not grouped
    // by engine (bit 1, k_vm_index) was cut into meaningless slices: nothing of this burst is kept -- the host's copy of the states is
    // still the one from before it -- and the host path replays it
    HIP_OK(hipMemcpyAsync(p->h_vm_ovf, p->d_vm_ovf, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_OK(hipEventRecord(p->ev_vm_events, st));
    HIP_OK(hipStreamSynchronize(st));
    if (*p->h_vm_ovf) { p->vm_host_dirty = 1; return false; }
    // the states come back on a stream of their own (126 MB for 131 072 engines, 2.3 ms), and the queues are applied AT ONCE beside that
    // (Voice::note_on builds its voice when it is called, voice.rs:28-110; the next render used to start with this launch): the 14.5 ms
    // of a whole-pool re-strike's k_apply_ops now cover the download and the host's packing, scanning and list building for the block
    // that follows.  Ops the host had queued before the burst keep it off this path altogether (above); ops it queues afterwards are
    // applied by the next render, behind these, as before.
    hipStream_t sc = p->pipe_stream[1] ? p->pipe_stream[1] : st;
    if (sc != st) HIP_OK(hipStreamWaitEvent(sc, p->ev_vm_events, 0));
    HIP_OK(hipMemcpyAsync(p->h_vm + e_lo, p->d_vm + e_lo, sizeof(OwVm) * (e_hi - e_lo), hipMemcpyDeviceToHost, sc));
    HIP_OK(hipEventRecord(p->ev_vm, sc));
    p->vm_download_pending = true;
    std::memset(p->dirty.data() + e_lo, 1, e_hi - e_lo);           // masks / queues of these engines changed: the next render packs them
    __atomic_store_n(&p->dirty_any, (uint8_t)1, __ATOMIC_RELAXED);
    // From here on the burst IS applied to the states (the download will overwrite the host's copy): whatever fails below, the caller must
    // not replay the list on the host.  A failed early application leaves the queues to the next render, as OW_MIDI_APPLY_EARLY=0 does.
    bool early = false, launched = false;
    if (p->sw.midi_apply_early && sc != st) {
        try {
            owdev::k_apply_ops<<<dim3(e_hi - e_lo), dim3(64), 0, st>>>(p->dK, p->d_nt, p->d_vrec, nullptr, nullptr, nullptr, p->d_ops_fix, p->d_vm, (int)e_lo);
            HIP_OK(hipGetLastError());
            launched = true;
            // the queue lengths go back to zero only when the download has them: the host recognises the engines of this burst by them
            // (vm_settle_applied); a short k_apply_ops -- a burst of note-offs -- would otherwise finish, and clear, under the copy
            HIP_OK(hipStreamWaitEvent(st, p->ev_vm, 0));
            owdev::k_vm_clear_dev_ops<<<dim3((e_hi - e_lo + 255) / 256), dim3(256), 0, st>>>(p->d_vm, e_lo, e_hi);
            HIP_OK(hipGetLastError());
            early = true;
        } catch (const std::exception& ex) {
            (void)hipGetLastError();
            set_err(std::string("ow_pool_midi (device burst, early application): ") + ex.what());
        }
    }
    if (early || launched) { p->dev_ops_applied = true; p->applied_lo = e_lo; p->applied_hi = e_hi; }
    else p->dev_ops_pending = true;
    p->vm_bursts += 1;
    return true;
}

void ow_pool_midi(ow_pool* p, const ow_midi_event* ev, size_t n) {
    if (!p || !ev) return;
    // Engines are independent state machines: large event lists are applied by the persistent host workers, each slice owning a
    // contiguous range of engines and walking the list in array order (per-engine order is what matters).  No allocation here.
    size_t T = std::min<size_t>(effective_cpus(), OW_MAX_SLICES);
    if (p->sw.midi_threads) T = (size_t)p->sw.midi_threads;
    if (n < 4096 || p->I < 2 * T) T = 1;
    vm_wait_download(p);
    const bool want_device = n > 0 && !p->voices_only && (p->sw.midi_device == 1 || (p->sw.midi_device < 0 && p->I >= 8192 && n >= 65536));
    if (want_device) {     // (whether the list is grouped by engine is checked on the device too: no walk over 200 MB of events here)
        bool done = false;
        guarded("ow_pool_midi (device burst)", [&] { done = midi_burst_on_device(p, ev, n); });
        if (done) return;
    }
    if (T == 1) {
        for (size_t i = 0; i < n; ++i) if (ev[i].engine < p->I) midi_apply_one(p, ev[i]);
        return;
    }
    // Fast path: a list grouped by engine (non-decreasing engine index, the usual layout of a batched script) is cut into T
    // contiguous slices at engine boundaries, so each slice touches only its own events.  The grouping is verified first, in
    // parallel; an ungrouped list falls back to every slice scanning the whole list for its engine range.
    uint8_t ok[OW_MAX_SLICES];
    auto verify = [&](size_t t) {
        const size_t i0 = std::max<size_t>(n * t / T, 1), i1 = n * (t + 1) / T;
        uint8_t good = 1;
        for (size_t i = i0; i < i1; ++i) good &= (uint8_t)(ev[i].engine >= ev[i - 1].engine);
        ok[t] = good;
    };
    Workers::get().each(T, verify);
    bool grouped = true;
    for (size_t t = 0; t < T; ++t) grouped = grouped && ok[t];
    if (grouped) {
        size_t cut[OW_MAX_SLICES + 1];
        cut[0] = 0; cut[T] = n;
        for (size_t t = 1; t < T; ++t) {   // first event of the engine that owns position n*t/T belongs to the slice on the right
            size_t i = n * t / T;
            const uint32_t eng = ev[i].engine;
            i = (size_t)(std::lower_bound(ev, ev + i, eng, [](const ow_midi_event& a, uint32_t b) { return a.engine < b; }) - ev);
            cut[t] = std::max(i, cut[t - 1]);
        }
        auto apply = [&](size_t t) {
            for (size_t i = cut[t]; i < cut[t + 1]; ++i) if (ev[i].engine < p->I) midi_apply_one(p, ev[i]);
        };
        Workers::get().each(T, apply);
    } else {
        const size_t per = (p->I + T - 1) / T;
        auto apply = [&](size_t t) {
            const uint32_t lo = (uint32_t)(t * per), hi = (uint32_t)std::min(p->I, (t + 1) * per);
            for (size_t i = 0; i < n; ++i) if (ev[i].engine >= lo && ev[i].engine < hi) midi_apply_one(p, ev[i]);
        };
        Workers::get().each(T, apply);
    }
}

// ---- host-logic test hooks (no device) -----------------------------------------------------------
ow_engine* ow_test_engine_new(double sample_rate) {
    ow_engine* e = new ow_engine();
    e->sr = sample_rate;
    return e;
}
void ow_test_engine_free(ow_engine* e) { if (e && !e->pool) delete e; }
size_t ow_test_engine_take_ops(ow_engine* e, uint8_t* type, uint8_t* slot, uint8_t* note, uint32_t* seed, double* velocity, size_t cap) {
    if (!e) return 0;
    const size_t n = std::min(cap, e->ops.size());
    for (size_t i = 0; i < n; ++i) {
        type[i] = e->ops[i].type; slot[i] = e->ops[i].slot; note[i] = e->ops[i].note; seed[i] = e->ops[i].seed; velocity[i] = e->ops[i].velocity;
    }
    const size_t total = e->ops.size();
    e->ops.clear();
    return total;
}
void ow_test_engine_after_render(ow_engine* e, size_t len, uint64_t silent_mask) {
    if (!e) return;
    OwEngineOut o;
    std::memset(&o, 0, sizeof o);
    o.silent_mask = silent_mask;
    engine_post_render(e, (uint32_t)std::min<size_t>(len, 0xFFFFFFFFull), o);
}
uint64_t ow_test_engine_masks(const ow_engine* e, int which) { return e ? (which ? e->vm->steal_mask : e->vm->main_mask) : 0; }
int ow_test_pool_stagger_tremolo(ow_pool* p, size_t n_groups) {
    if (!p || n_groups == 0 || n_groups > p->I) return -1;
    try {
        HIP_OK(hipSetDevice(p->device));
        invalidate_spec(p);
        HIP_OK(hipStreamSynchronize(p->stream));
        const uint32_t I = (uint32_t)p->I, G = (uint32_t)n_groups;
        const long long period0 = (long long)(p->hc.os_sr / 5.6);
        const long long step0 = std::max<long long>(1, period0 / (long long)G);
        if (p->traj) {
            // on the shared trajectory "group g runs g * step samples ahead" is a shift of its engines' births (from the pool's common t)
            if (p->n_on_traj != p->I) throw std::runtime_error("stagger: an engine has left the trajectory");
            const long long b0 = p->min_birth;
            for (uint32_t e = 0; e < I; ++e) p->h_birth[e] = b0 - (long long)(e % G) * step0;
            traj_upload_births(p, 0, (int)I);
            traj_recount(p);
            return 0;
        }
        // every engine takes the oscillator state of its current leader, then group g = {g, g + G, ...} is led by engine g
        size_t n_copy = 0;
        for (uint32_t g = 0; g < G; ++g) { p->h_copy[n_copy] = p->h_lead[g]; p->h_copy[I + n_copy] = g; ++n_copy; }
        HIP_OK(hipMemcpyAsync(p->d_copy, p->h_copy, sizeof(uint32_t) * n_copy, hipMemcpyHostToDevice, p->stream));
        HIP_OK(hipMemcpyAsync(p->d_copy + I, p->h_copy + I, sizeof(uint32_t) * n_copy, hipMemcpyHostToDevice, p->stream));
        // leaders of the old groups are among the sources: copy through the backup buffer so that no source row is overwritten first
        HIP_OK(hipMemcpyAsync(p->d_trem_backup, p->d_cs, sizeof(double) * 18 * p->I, hipMemcpyDeviceToDevice, p->stream));
        owdev::k_trem_copy_rows_from<<<dim3((unsigned)((n_copy + 63) / 64)), dim3(64), 0, p->stream>>>(p->d_cs, p->d_trem_backup, (int)I, p->d_copy, p->d_copy + I, (int)n_copy);
        for (uint32_t e = 0; e < I; ++e) p->h_lead[e] = e % G;
        trem_groups_changed(p);
        trem_leader_list(p, 0, (int)I);
        // group g runs g * step samples ahead of group 0; the steps cover one period of the ~5.6 Hz oscillator
        const long long period = (long long)(p->hc.os_sr / 5.6);
        const long long step = std::max<long long>(1, period / (long long)G);
        if (p->tremolo_kind == OW_TREMOLO_LEGACY_LFO)
            owdev::k_tremolo_lfo<<<dim3((p->n_lead + 63) / 64), dim3(64), 0, p->stream>>>(p->dK, p->d_cs, nullptr, (int)I, 0LL, p->d_leaders, p->n_lead, step);
        else
            owdev::k_trem_settle<<<dim3((p->n_lead + 63) / 64), dim3(64), 0, p->stream>>>(p->dK, p->d_cs, (int)I, p->d_leaders, p->n_lead, 0LL, step);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(p->stream));
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_test_pool_stagger_tremolo: ") + ex.what()); return -1; }
}
size_t ow_test_pool_tremolo_groups(const ow_pool* p) {
    if (!p) return 0;
    size_t n = 0;
    if (p->traj) {       // distinct tremolo phases: engines on the trajectory share an oscillator exactly when they stand at the same t
        std::vector<long long> b;
        for (size_t e = 0; e < p->I; ++e) { if (p->h_birth[e] != OW_OFF_TRAJ) b.push_back(p->h_birth[e]); else n += p->h_lead[e] == (uint32_t)e; }
        std::sort(b.begin(), b.end());
        return n + (size_t)(std::unique(b.begin(), b.end()) - b.begin());
    }
    for (size_t e = 0; e < p->I; ++e) n += p->h_lead[e] == (uint32_t)e;
    return n;
}
int ow_debug_power_amp(double sample_rate, const double* in, size_t n_rows, size_t n, int rail_sag, const long long* poke_at, const int* poke_node,
                       const double* poke_val, double* out, double* taps, int device) {
    try {
        if (!in || !out || n_rows == 0 || n == 0 || !(sample_rate > 0.0)) throw std::runtime_error("bad argument");
        HIP_OK(hipSetDevice(device));
        StreamOwner so;
        HIP_OK(hipStreamCreateWithFlags(&so.s, hipStreamNonBlocking));
        std::unique_ptr<OwPaConsts> hc(new OwPaConsts());
        owhip::build_pa_consts(*hc, sample_rate);
        DevMem dC, dS, dIn, dOut, dT, dPa, dPn, dPv;
        dC.alloc(sizeof(OwPaConsts)); dS.alloc(sizeof(double) * owdev::PAS_CIRCUIT_END);
        dIn.alloc(sizeof(double) * n_rows * n); dOut.alloc(sizeof(double) * n_rows * n);
        if (taps) dT.alloc(sizeof(double) * n_rows * n * 3);
        pa_settled_to_device(device, dS.as<double>(), so.s);
        HIP_OK(hipMemcpyAsync(dC.p, hc.get(), sizeof(OwPaConsts), hipMemcpyHostToDevice, so.s));
        HIP_OK(hipMemcpyAsync(dIn.p, in, sizeof(double) * n_rows * n, hipMemcpyHostToDevice, so.s));
        if (poke_at) {
            dPa.alloc(sizeof(long long) * n_rows); dPn.alloc(sizeof(int) * n_rows); dPv.alloc(sizeof(double) * n_rows);
            HIP_OK(hipMemcpyAsync(dPa.p, poke_at, sizeof(long long) * n_rows, hipMemcpyHostToDevice, so.s));
            HIP_OK(hipMemcpyAsync(dPn.p, poke_node, sizeof(int) * n_rows, hipMemcpyHostToDevice, so.s));
            HIP_OK(hipMemcpyAsync(dPv.p, poke_val, sizeof(double) * n_rows, hipMemcpyHostToDevice, so.s));
        }
        owdev::k_mpa_debug<<<dim3((unsigned)((n_rows + PA_EPB - 1) / PA_EPB)), dim3(PA_WPB * 64), 0, so.s>>>(dC.as<OwPaConsts>(), dS.as<double>(), dIn.as<double>(), dOut.as<double>(),
                                                                          taps ? dT.as<double>() : nullptr, (long long)n, (int)n_rows, rail_sag,
                                                                          poke_at ? dPa.as<long long>() : nullptr, dPn.as<int>(), dPv.as<double>());
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpyAsync(out, dOut.p, sizeof(double) * n_rows * n, hipMemcpyDeviceToHost, so.s));
        if (taps) HIP_OK(hipMemcpyAsync(taps, dT.p, sizeof(double) * n_rows * n * 3, hipMemcpyDeviceToHost, so.s));
        HIP_OK(hipStreamSynchronize(so.s));
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_debug_power_amp: ") + ex.what()); return -1; }
}
int ow_test_pool_power_amp_passes(ow_pool* p, uint32_t* out, size_t n) {
    if (!p || !out || !p->d_pa_demand || n != (size_t)p->I) return -1;
    if (hipSetDevice(p->device) != hipSuccess) return -1;
    if (hipStreamSynchronize(p->stream) != hipSuccess) return -1;
    return hipMemcpy(out, p->d_pa_demand, sizeof(uint32_t) * n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
int ow_test_pool_enable_power_amp_tap(ow_pool* p) {
    if (!p || p->power_amp_kind != OW_POWER_AMP_MELANGE) return -1;
    if (p->d_pa_tap) return 0;
    if (hipSetDevice(p->device) != hipSuccess || hipStreamSynchronize(p->stream) != hipSuccess) return -1;
    return hipMalloc(&p->d_pa_tap, sizeof(double) * 2 * p->Lcap * p->I) == hipSuccess ? 0 : -1;
}
int ow_test_pool_read_power_amp_out(ow_pool* p, double* out_host, size_t out_stride, size_t n_os) {
    if (!p || !out_host || !p->d_pa_tap || n_os > 2 * p->Lcap) return -1;
    try {
        HIP_OK(hipSetDevice(p->device));
        const size_t I = p->I;
        std::vector<double> a(I * n_os);
        HIP_OK(hipMemcpy(a.data(), p->d_pa_tap, sizeof(double) * I * n_os, hipMemcpyDeviceToHost));  // [n_os][I]
        for (size_t e = 0; e < I; ++e)
            for (size_t n = 0; n < n_os; ++n) out_host[e * out_stride + n] = a[n * I + e];
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_test_pool_read_power_amp_out: ") + ex.what()); return -1; }
}
int ow_test_engine_poke_power_amp_node(ow_engine* e, int node, double volts) {
    if (!e || !e->pool || e->pool->power_amp_kind != OW_POWER_AMP_MELANGE || node < 0 || node >= PA_N) return -1;
    ow_pool* p = e->pool;
    if (hipSetDevice(p->device) != hipSuccess || hipStreamSynchronize(p->stream) != hipSuccess) return -1;
    return hipMemcpy(p->d_pa + (size_t)(owdev::PAS_V + node) * p->I + e->index, &volts, sizeof volts, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
// Host matrix builders (ow_consts_host.hpp) of the three generated solvers at `rate` (the solver's own rate: the chain rate), no device.
// force_rebuild != 0 bypasses the "codegen rate -> baked tables" shortcut so that the rebuild can be checked against those tables.
int ow_test_host_matrices(int solver, double rate, int force_rebuild, double* s, double* k, double* sni, double* aneg,
                          double* s_be, double* k_be, double* sni_be, double* aneg_be) {
    try {
        if (!(rate > 0.0)) throw std::runtime_error("bad rate");
        auto put = [](double* dst, const void* src, size_t n) { if (dst) std::memcpy(dst, src, sizeof(double) * n); };
        struct Force { Force(bool on) { owhip::g_force_rebuild = on; } ~Force() { owhip::g_force_rebuild = false; } } force(force_rebuild != 0);
        if (solver == 0 || solver == 1) {
            std::unique_ptr<OwConsts> c(new OwConsts());
            // build_consts takes the HOST rate; a host rate >= 88.2 kHz runs the chain at that rate without oversampling (engine.rs:195)
            owhip::build_consts(*c, rate < 88200.0 ? rate * 0.5 : rate, solver == 1 ? OW_PREAMP_MELANGE12 : OW_PREAMP_LEGACY8);
            if (c->os_sr != rate) throw std::runtime_error("rate is not reachable as a chain rate");
            if (solver == 0) {
                put(s, c->t_s, 49); put(k, c->t_k, 16); put(sni, c->t_s_ni, 28); put(aneg, c->t_a_neg, 49);
                put(s_be, c->t_s_be, 49); put(k_be, c->t_k_be, 16); put(sni_be, c->t_s_ni_be, 28); put(aneg_be, c->t_a_neg_be, 49);
                return 704;
            }
            put(s, c->m_s0, 144); put(k, c->m_k0, 9); put(sni, c->m_sni0, 36); put(aneg, c->m_aneg0, 144);
            // the melange preamp's backward-Euler set is never rebuilt (gen_preamp.rs:2058-2061): the kernels read the baked tables
            put(s_be, PRE_S_BE_DEFAULT, 144); put(k_be, PRE_K_BE_DEFAULT, 9); put(sni_be, PRE_S_NI_BE_DEFAULT, 36); put(aneg_be, PRE_A_NEG_BE_DEFAULT, 144);
            return 1203;
        }
        if (solver == 2) {
            std::unique_ptr<OwPaConsts> c(new OwPaConsts());
            owhip::build_pa_consts(*c, rate);
            put(s, c->s, 400); put(k, c->k, 256); put(sni, c->s_ni, 320); put(aneg, c->a_neg, 400);
            put(s_be, c->s_be, 400); put(k_be, c->k_be, 256); put(sni_be, c->s_ni_be, 320); put(aneg_be, c->a_neg_be, 400);
            return 2016;
        }
        throw std::runtime_error("unknown solver");
    } catch (const std::exception& ex) { set_err(std::string("ow_test_host_matrices: ") + ex.what()); return -1; }
}
// Forget the process-wide settled states (Twin-T per chain rate; melange preamp and power amp per device): the next pool settles
// afresh on the device.  Returns the number of cached Twin-T states that were dropped.
int ow_test_clear_settle_caches(void) {
    int n;
    {   // one lock at a time: traj_acquire holds g_traj_mu while its settle takes g_mel_mu (trem_settled_rows)
        std::lock_guard<std::mutex> lk(g_mel_mu);
        n = (int)g_trem_settled.size();
        g_trem_settled.clear(); g_mel_settled.clear(); g_pa_settled.clear();
    }
    { std::lock_guard<std::mutex> lt(g_traj_mu); traj_registry().clear(); }     // pools that hold a store keep it alive; new pools start a new one
    return n;
}
// Overwrite one field of a voice record (VF_* of ow_types.h) on the device before the next block: the way to make a voice non-finite,
// which no API call can (voice-sum NaN guard, engine.rs:496-521).
static_assert(VF_Q == OW_TEST_VF_Q && VF_S == OW_TEST_VF_S0, "openwurli_hip_test.h names two voice-record fields by index");
int ow_test_engine_poke_voice(ow_engine* e, int slot, int steal, int field, double value) {
    if (!e || !e->pool || slot < 0 || slot >= OW_MAX_VOICES || field < 0 || field >= VF_COUNT) return -1;
    ow_pool* p = e->pool;
    if (hipSetDevice(p->device) != hipSuccess || hipStreamSynchronize(p->stream) != hipSuccess) return -1;
    double* dst = p->d_vrec + ((size_t)e->index * 2 + (steal ? 1 : 0)) * OW_VREC_DOUBLES + (size_t)field * 64 + slot;
    return hipMemcpy(dst, &value, sizeof value, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
// plain device-to-host copy (tests read blocks that a render left in HBM: ow_pool_device_output)
int ow_test_device_read(void* dst_host, const void* src_device, size_t bytes, int device) {
    if (!dst_host || !src_device) return -1;
    if (hipSetDevice(device) != hipSuccess) return -1;
    return hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#ifdef OW_DBG_COUNTERS
// development counters (built with OW_HIPCC_EXTRA=-DOW_DBG_COUNTERS only): read and clear
extern "C" int ow_debug_counters(unsigned long long* out8, int device) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(owdev::g_ow_dbg), sizeof z) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(owdev::g_ow_dbg), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
// The two oscillator kernels of the shared trajectory on their own (no store, no pool): CircuitState at DC_OP, n_settle steps without
// the cell (Tremolo::new's settle), then n steps with it through a trajectory kernel in launches of `chunk` steps.  row = 0: the quad-lane
// kernels (k_tremolo_wide<true>, k_trem_traj_extend), 1: the row kernels (ow_trem_row.h).  r_out[n], state_out[18], ckpt_out[(n / 4096 + 2)
// * 16] (or NULL), be_out[1 + 1023] (or NULL); *ms_out = device time of the trajectory launches (HIP events).
int ow_debug_trem_trajectory(double sample_rate, long long n_settle, long long n, long long chunk, int row, double* r_out, double* state_out,
                             double* ckpt_out, unsigned long long* be_out, double* ms_out, int device) {
    try {
        if (!r_out || !state_out || n <= 0 || n_settle < 0 || chunk <= 0) throw std::runtime_error("bad arguments");
        HIP_OK(hipSetDevice(device));
        std::unique_ptr<OwConsts> hc(new OwConsts());
        owhip::build_consts(*hc, sample_rate, OW_PREAMP_LEGACY8);
        DevMem dk, dstate, dr, dck, dbe, dzero;
        dk.alloc(sizeof(OwConsts)); dstate.alloc(sizeof(double) * 18); dr.alloc(sizeof(double) * (size_t)(n + 64));
        const size_t nck = TremTraj::ckpt_doubles((size_t)n);
        dck.alloc(sizeof(double) * nck); dbe.alloc(sizeof(unsigned long long) * (1 + OW_TRAJ_BE_CAP)); dzero.alloc(sizeof(uint32_t));
        StreamOwner so;
        HIP_OK(hipStreamCreate(&so.s));
        HIP_OK(hipMemcpyAsync(dk.p, hc.get(), sizeof(OwConsts), hipMemcpyHostToDevice, so.s));
        HIP_OK(hipMemsetAsync(dck.p, 0, sizeof(double) * nck, so.s));
        HIP_OK(hipMemsetAsync(dbe.p, 0xFF, sizeof(unsigned long long) * (1 + OW_TRAJ_BE_CAP), so.s));
        HIP_OK(hipMemsetAsync(dbe.p, 0, sizeof(unsigned long long), so.s));
        HIP_OK(hipMemsetAsync(dzero.p, 0, sizeof(uint32_t), so.s));
        owdev::k_trem_state_dc<<<dim3(1), dim3(64), 0, so.s>>>(dstate.as<double>());
        if (n_settle > 0) {
            if (row) owdev::k_trem_settle_row<<<dim3(1), dim3(64), 0, so.s>>>(dk.as<OwConsts>(), dstate.as<double>(), n_settle);
            else owdev::k_tremolo_wide<true><<<dim3(1), dim3(64), 0, so.s>>>(dk.as<OwConsts>(), dstate.as<double>(), nullptr, 1, n_settle, dzero.as<uint32_t>(), 1);
        }
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
        HIP_OK(hipEventRecord(e0, so.s));
        for (long long t = 0; t < n; t += chunk) {
            const long long m = std::min(chunk, n - t);
            if (row) owdev::k_trem_traj_extend_row<<<dim3(1), dim3(64), 0, so.s>>>(dk.as<OwConsts>(), dstate.as<double>(), dr.as<double>() + t, t, m, dck.as<double>(), dbe.as<unsigned long long>());
            else owdev::k_trem_traj_extend<<<dim3(1), dim3(64), 0, so.s>>>(dk.as<OwConsts>(), dstate.as<double>(), dr.as<double>() + t, t, m, dck.as<double>(), dbe.as<unsigned long long>());
        }
        HIP_OK(hipEventRecord(e1, so.s));
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(so.s));
        float ms = 0.0f;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        hipEventDestroy(e0); hipEventDestroy(e1);
        if (ms_out) *ms_out = ms;
        HIP_OK(hipMemcpy(r_out, dr.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(state_out, dstate.p, sizeof(double) * 18, hipMemcpyDeviceToHost));
        if (ckpt_out) HIP_OK(hipMemcpy(ckpt_out, dck.p, sizeof(double) * nck, hipMemcpyDeviceToHost));
        if (be_out) HIP_OK(hipMemcpy(be_out, dbe.p, sizeof(unsigned long long) * (1 + OW_TRAJ_BE_CAP), hipMemcpyDeviceToHost));
        return 0;
    } catch (const std::exception& ex) { (void)hipGetLastError(); set_err(std::string("ow_debug_trem_trajectory: ") + ex.what()); return -1; }
}
// Which literal-rebuild fast paths the host found usable for the melange preamp at chain rate `rate` (no device): bit 0 = the
// R-independent leading block could be replayed (ml_ok), bit 1 = the factors have the sparsity pattern ow_melange_col.h compiles in.
int ow_test_host_melange_paths(double rate) {
    try {
        std::unique_ptr<OwConsts> c(new OwConsts());
        owhip::build_consts(*c, rate < 88200.0 ? rate * 0.5 : rate, OW_PREAMP_MELANGE12);
        if (c->os_sr != rate) throw std::runtime_error("rate is not reachable as a chain rate");
        return (c->ml_ok ? 1 : 0) | (c->ml_sparse_ok ? 2 : 0);
    } catch (const std::exception& ex) { set_err(std::string("ow_test_host_melange_paths: ") + ex.what()); return -1; }
}
void ow_test_inject_render_faults(ow_pool* p, int n_renders) { if (p) p->inject_faults = n_renders > 0 ? n_renders : 0; }
// One latched switch of a live pool (struct Switches; the OW_* environment variables are only read when a pool is created).
int ow_test_pool_set_switch(ow_pool* p, const char* name, int value) {
    if (!p || !name) return -1;
    const std::string n(name);
    if (hipSetDevice(p->device) != hipSuccess) return -1;
    invalidate_spec(p);                                   // a pending block-ahead result was produced under the old schedule
    if (hipStreamSynchronize(p->stream) != hipSuccess) return -1;
    Switches& w = p->sw;
    if (n == "trem_serial") w.trem_serial = value != 0;
    else if (n == "trem_wide") w.trem_wide = value < 0 ? -1 : (value != 0);
    else if (n == "preamp_wide") w.preamp_wide = value < 0 ? -1 : (value != 0);
    else if (n == "chain_fused") w.chain_fused = value < 0 ? -1 : (value != 0);
    else if (n == "mel_generic") w.mel_generic = value != 0;
    else if (n == "mel_rank1") w.mel_rank1 = value != 0;
    else if (n == "mel_lds") w.mel_lds = value != 0;
    else if (n == "mel_eng") w.mel_eng = value != 0;
    else if (n == "eout_attn") w.eout_attn = value < 0 ? -1 : (value != 0);
    else if (n == "voice_skew") w.voice_skew = value != 0;
    else if (n == "pa_sort") { if (value < 0 || value > 2) return -1; w.pa_sort = value; }
    else if (n == "host_profile") w.host_profile = value != 0;
    else if (n == "out_direct") w.out_direct = value < 0 ? -1 : (value != 0);
    else if (n == "midi_device") w.midi_device = value < 0 ? -1 : (value != 0);
    else if (n == "voice_attack") { w.voice_attack = value != 0; p->lists_valid = false; }
    else if (n == "voice_steal") w.voice_steal = value != 0;
    else if (n == "voice_release") w.voice_release = value != 0;
    else if (n == "midi_apply_early") w.midi_apply_early = value != 0;
    else if (n == "force_general") { w.force_general = value != 0; p->lists_valid = false; }
    else if (n == "chain_stream") w.chain_stream = value < 0 ? -1 : (value != 0);
    else if (n == "chain_row") w.chain_row = value < 0 ? -1 : (value != 0);
    else return -1;                                       // (trem_traj / trem_cache / pipe shape the pool at creation: environment only)
    return 0;
}
int ow_test_pool_get_switch(const ow_pool* p, const char* name) {
    if (!p || !name) return -2;
    const std::string n(name);
    const Switches& w = p->sw;
    if (n == "trem_serial") return w.trem_serial;
    if (n == "trem_wide") return w.trem_wide;
    if (n == "preamp_wide") return w.preamp_wide;
    if (n == "chain_fused") return w.chain_fused;
    if (n == "chain_row") return w.chain_row;
    if (n == "mel_generic") return w.mel_generic;
    if (n == "mel_rank1") return w.mel_rank1;
    if (n == "mel_lds") return w.mel_lds;
    if (n == "mel_eng") return w.mel_eng;
    if (n == "eout_attn") return w.eout_attn;
    if (n == "voice_skew") return w.voice_skew;
    if (n == "voice_skew_active") return p->skew_next ? 1 : 0;   // the next steady launch takes the skewed variant
    if (n == "pa_sort") return w.pa_sort;
    if (n == "trem_traj") return p->traj ? 1 : 0;
    if (n == "trem_cache") return w.trem_cache;
    if (n == "out_direct") return w.out_direct;
    if (n == "midi_device") return w.midi_device;
    if (n == "voice_attack") return w.voice_attack;
    if (n == "voice_steal") return w.voice_steal;
    if (n == "voice_release") return w.voice_release;
    if (n == "midi_apply_early") return w.midi_apply_early;
    if (n == "blocks_steady") return (int)p->vl_steady.n_blocks;   // wavefront blocks of the voice lists the last render launched
    if (n == "blocks_general") return (int)p->vl_general.n_blocks;
    if (n == "blocks_attack") return (int)p->vl_attack.n_blocks;
    if (n == "blocks_steal") return (int)p->vl_steal.n_blocks;
    if (n == "midi_device_bursts") return (int)std::min<uint64_t>(p->vm_bursts, 0x7FFFFFFF);
    if (n == "chain_stream") return w.chain_stream;
    return -2;
}
// Engines of the pool that read the shared trajectory / samples the store of the pool's rate holds (produced or enqueued) / its capacity.
int ow_test_pool_trajectory_info(const ow_pool* p, uint64_t out[3]) {
    if (!p || !out) return -1;
    out[0] = p->traj ? p->n_on_traj : 0; out[1] = 0; out[2] = 0;
    if (p->traj) { std::lock_guard<std::mutex> lk(p->traj->mu); out[1] = p->traj->len; out[2] = p->traj->cap_max; }
    return 0;
}
// The store behind the pool, in samples: [0] produced or enqueued, [1] known complete (finished launches), [2] what its buffers hold now,
// [3] its configured capacity, [4] t of the pool's oldest engine on it (what the next block needs is [4] + its chain-rate samples).
int ow_test_pool_trajectory_state(const ow_pool* p, uint64_t out[5]) {
    if (!p || !out) return -1;
    for (int i = 0; i < 5; ++i) out[i] = 0;
    if (!p->traj) return 0;
    std::lock_guard<std::mutex> lk(p->traj->mu);
    p->traj->in_flight();                                  // looks at the marks: refreshes `done`
    out[0] = p->traj->len; out[1] = p->traj->done; out[2] = p->traj->cap; out[3] = p->traj->cap_max;
    out[4] = (uint64_t)std::max<long long>(p->trem_clock - p->min_birth, 0);
    return 0;
}

// ---- diagnostics ---------------------------------------------------------------------------------
int ow_debug_mlp_raw(const uint8_t* notes, const double* velocities, size_t n, double* out, int use_mfma, int device) {
    try {
        if (!notes || !velocities || !out || n == 0) throw std::runtime_error("null argument");
        HIP_OK(hipSetDevice(device));
        DevMem dn, dv, dout;
        dn.alloc(n); dv.alloc(n * sizeof(double)); dout.alloc(n * 11 * sizeof(double));
        HIP_OK(hipMemcpy(dn.p, notes, n, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dv.p, velocities, n * sizeof(double), hipMemcpyHostToDevice));
        owdev::k_debug_mlp<<<dim3((unsigned)((n + 63) / 64)), dim3(64)>>>(dn.as<uint8_t>(), dv.as<double>(), (int)n, dout.as<double>(), use_mfma);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(out, dout.p, n * 11 * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_debug_mlp_raw: ") + ex.what()); return -1; }
}

int ow_debug_div(const double* a, const double* b, size_t n, double* fast, double* ieee, int device) {
    try {
        if (!a || !b || !fast || !ieee) throw std::runtime_error("null argument");
        if (n == 0) return 0;
        HIP_OK(hipSetDevice(device));
        DevMem da, db, df, di;
        da.alloc(n * sizeof(double)); db.alloc(n * sizeof(double)); df.alloc(n * sizeof(double)); di.alloc(n * sizeof(double));
        HIP_OK(hipMemcpy(da.p, a, n * sizeof(double), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(db.p, b, n * sizeof(double), hipMemcpyHostToDevice));
        owdev::k_debug_div<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(da.as<double>(), db.as<double>(), n, df.as<double>(), di.as<double>());
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(fast, df.p, n * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(ieee, di.p, n * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_debug_div: ") + ex.what()); return -1; }
}

int ow_debug_div_const(int which, const double* a, size_t n, double* fast, double* ieee, uint64_t* mismatches, int device) {
    try {
        if (which < 0 || which > 7) throw std::runtime_error("unknown constant");
        HIP_OK(hipSetDevice(device));
        if (!a) {
            if (!mismatches) throw std::runtime_error("null argument");
            if (which != 0 && which != 6) throw std::runtime_error("no exhaustive numerator set for this constant");
            DevMem dm;
            dm.alloc(sizeof(unsigned long long));
            HIP_OK(hipMemset(dm.p, 0, sizeof(unsigned long long)));
            owdev::k_debug_div_draw_all<<<dim3(4096), dim3(256)>>>(which, dm.as<unsigned long long>());
            HIP_OK(hipGetLastError());
            unsigned long long h = 0;
            HIP_OK(hipMemcpy(&h, dm.p, sizeof h, hipMemcpyDeviceToHost));
            *mismatches = h;
            return 0;
        }
        if (!fast || !ieee) throw std::runtime_error("null argument");
        if (n == 0) return 0;
        DevMem da, df, di;
        da.alloc(n * sizeof(double)); df.alloc(n * sizeof(double)); di.alloc(n * sizeof(double));
        HIP_OK(hipMemcpy(da.p, a, n * sizeof(double), hipMemcpyHostToDevice));
        owdev::k_debug_div_const<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(which, da.as<double>(), n, df.as<double>(), di.as<double>());
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(fast, df.p, n * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(ieee, di.p, n * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_debug_div_const: ") + ex.what()); return -1; }
}

int ow_debug_div_forms(int mode, const double* a, const double* b, const double* y, size_t n, double* fast, double* ieee, int device) {
    try {
        if (!a || !b || !fast || !ieee || mode < 0 || mode > 2 || (mode == 0 && !y)) throw std::runtime_error("bad argument");
        if (n == 0) return 0;
        HIP_OK(hipSetDevice(device));
        DevMem da, db, dy, df, di;
        da.alloc(n * sizeof(double)); db.alloc(n * sizeof(double)); dy.alloc(n * sizeof(double)); df.alloc(n * sizeof(double)); di.alloc(n * sizeof(double));
        HIP_OK(hipMemcpy(da.p, a, n * sizeof(double), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(db.p, b, n * sizeof(double), hipMemcpyHostToDevice));
        if (y) HIP_OK(hipMemcpy(dy.p, y, n * sizeof(double), hipMemcpyHostToDevice));
        owdev::k_debug_div_forms<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(mode, da.as<double>(), db.as<double>(), dy.as<double>(), n, df.as<double>(), di.as<double>());
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(fast, df.p, n * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(ieee, di.p, n * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_debug_div_forms: ") + ex.what()); return -1; }
}

int ow_debug_unary(int which, const double* x, size_t n, double* fast, double* lib, int device) {
    try {
        if (!x || !fast || !lib || which < 0 || which > 6) throw std::runtime_error("null argument or unknown function");
        if (n == 0) return 0;
        HIP_OK(hipSetDevice(device));
        DevMem dx, df, dl;
        dx.alloc(n * sizeof(double)); df.alloc(n * sizeof(double)); dl.alloc(n * sizeof(double));
        HIP_OK(hipMemcpy(dx.p, x, n * sizeof(double), hipMemcpyHostToDevice));
        owdev::k_debug_exp<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(which, dx.as<double>(), n, df.as<double>(), dl.as<double>());
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpy(fast, df.p, n * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(lib, dl.p, n * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_debug_unary: ") + ex.what()); return -1; }
}

// ---- offline ------------------------------------------------------------------------------------
static long long render_note_impl(uint8_t midi, double velocity, double dur_s, double sample_rate, int device, double* out, size_t cap, const double* ds);
long long ow_render_note(uint8_t midi, double velocity, double dur_s, double sample_rate, int device, double* out, size_t cap) {
    return render_note_impl(midi, velocity, dur_s, sample_rate, device, out, cap, nullptr);
}
long long ow_render_note_with_scale(uint8_t midi, double velocity, double dur_s, double sample_rate, double displacement_scale, int device,
                                    double* out, size_t cap) {
    return render_note_impl(midi, velocity, dur_s, sample_rate, device, out, cap, &displacement_scale);
}
double ow_normalize_scale(const double* samples, size_t n) {     // main.rs:505-511
    double peak = 0.0;
    for (size_t i = 0; samples && i < n; ++i) peak = std::fmax(peak, std::fabs(samples[i]));
    return peak > 0.7 ? 0.7 / peak : 1.0;
}
static long long render_note_impl(uint8_t midi, double velocity, double dur_s, double sample_rate, int device, double* out, size_t cap, const double* ds) {
    try {
        // a voices-only pool: no chain state, no Twin-T settle, and render_range launches the voice kernels alone
        struct PoolGuard { ow_pool* p; ~PoolGuard() { pool_destroy(p); } } guard{pool_create(sample_rate, 1, device, OW_PREAMP_LEGACY8, OW_POWER_AMP_BEHAVIORAL,
                                                                                               OW_TREMOLO_TWIN_T, true)};
        ow_pool* p = guard.p;
        ow_engine* e = p->engines[0];
        // Voice::render_note: seed = midi * 2654435761, MLP off, no note clamping beyond the table range (voice.rs:206-207)
        const uint8_t note = std::min<uint8_t>(std::max<uint8_t>(midi, OW_MIDI_LO), OW_MIDI_HI);
        e->vm->has_voice |= 1ull; e->set_state(0, OW_VOICE_HELD); e->vm->midi_of[0] = note;
        e->sync_masks(0);
        push_op(e, OP_NOTE_ON, 0, note, false, (uint32_t)midi * 2654435761u, velocity);
        if (ds) push_op(e, OP_SET_DS, 0, note, false, 0, *ds);      // voice.set_displacement_scale(scale) right after note_on (voice.rs:210-212)
        double x = dur_s * sample_rate;
        const size_t n = (!(x == x) || x <= 0.0) ? 0 : (size_t)x;
        const size_t chunk_len = p->Lcap;            // OW_MAX_BLOCK for a pool of one: few launches, few synchronisations
        std::vector<double> chunk(chunk_len);
        size_t done = 0;
        while (done < n) {
            const size_t len = std::min<size_t>(chunk_len, n - done);
            render_range(p, 0, 1, len, true);
            HIP_OK(hipStreamSynchronize(p->stream));
            // no cleanup_voices here: render_note keeps rendering the voice for the whole duration; only the kernel choice of the
            // next chunk follows the device status (onset ramp / attack noise still running -> general kernel again)
            {
                const uint8_t tr = p->h_eout[0].transient != 0u;
                if (p->h_eout[0].transient == 2u) throw std::runtime_error("voice dispatch: a voice in a transient phase was sent to the steady kernel");
                if (tr != p->transient[0]) { p->transient[0] = tr; p->lists_valid = false; }
            }
            if (ow_pool_read_voice_sum(p, chunk.data(), len, len) != 0) throw std::runtime_error(g_err);
            for (size_t i = 0; i < len && done + i < cap; ++i) out[done + i] = chunk[i];
            done += len;
        }
        return (long long)n;      // the note's length; min(n, cap) samples were written (header contract)
    } catch (const std::exception& ex) { set_err(std::string("ow_render_note: ") + ex.what()); return -1; }
}

long long ow_batch_render(const ow_job* jobs, size_t n_jobs, const ow_batch_cfg* cfg, double* out, size_t stride, int out_is_device) {
    try {
        if (!jobs || !cfg || !out || n_jobs == 0) throw std::runtime_error("null argument");
        if (cfg->struct_size != sizeof(ow_batch_cfg) || cfg->job_size != sizeof(ow_job))
            throw std::runtime_error("ABI mismatch: ow_batch_cfg.struct_size / job_size do not match this library's openwurli_hip.h (OW_ABI_VERSION " +
                                     std::to_string(OW_ABI_VERSION) + ")");
        if (cfg->preamp_kind != OW_PREAMP_LEGACY8 && cfg->preamp_kind != OW_PREAMP_MELANGE12) throw std::runtime_error("unknown preamp_kind");
        const double x = cfg->duration_s * cfg->sample_rate;
        const size_t n = (!(x == x) || x <= 0.0) ? 0 : (size_t)x;                 // (duration * sample_rate) as usize, main.rs:411
        if (n == 0) return 0;
        if (stride < n) throw std::runtime_error("stride smaller than the job length");
        int ndev = 0;
        HIP_OK(hipGetDeviceCount(&ndev));
        if (ndev <= 0) throw std::runtime_error("no HIP device: openwurli-hip has no CPU fallback");
        HIP_OK(hipSetDevice(cfg->device));
        OwConsts hc;
        owhip::build_consts(hc, cfg->sample_rate, cfg->preamp_kind);
        std::vector<owdev::OwJobDev> hj(n_jobs);
        for (size_t i = 0; i < n_jobs; ++i) {
            hj[i].note = jobs[i].note; hj[i].velocity = jobs[i].velocity; hj[i].mlp = jobs[i].mlp; hj[i].poweramp = jobs[i].poweramp;
            hj[i].no_preamp = jobs[i].no_preamp ? 1 : 0; hj[i].no_attack_noise = jobs[i].no_attack_noise ? 1 : 0;
            hj[i].has_ds = jobs[i].has_displacement_scale ? 1 : 0; hj[i].pad8 = 0;
            hj[i].volume = jobs[i].volume; hj[i].speaker = jobs[i].speaker; hj[i].r_ldr = jobs[i].r_ldr;
            hj[i].tremolo_depth = jobs[i].tremolo_depth; hj[i].displacement_scale = jobs[i].displacement_scale;
        }
        const size_t vblocks = (n_jobs + 63) / 64;
        StreamOwner so;
        HIP_OK(hipStreamCreateWithFlags(&so.s, hipStreamNonBlocking));
        hipStream_t st = so.s;
        DevMem m_K, m_nt, m_vrec, m_jobs, m_reed, m_out;   // released on every exit path
        m_K.alloc(sizeof(OwConsts));
        m_nt.alloc(sizeof(double) * NT_COUNT * 64);
        m_vrec.alloc(sizeof(double) * vblocks * OW_VREC_DOUBLES);
        m_jobs.alloc(sizeof(owdev::OwJobDev) * n_jobs);
        m_reed.alloc(sizeof(double) * n_jobs * stride);      // same row stride as the output: the chain kernels index both with it
        if (!out_is_device) m_out.alloc(sizeof(double) * n_jobs * stride);
        OwConsts* dK = m_K.as<OwConsts>(); double* d_nt = m_nt.as<double>(); double* d_vrec = m_vrec.as<double>();
        owdev::OwJobDev* d_jobs = m_jobs.as<owdev::OwJobDev>(); double* d_reed = m_reed.as<double>();
        double* d_out = out_is_device ? out : m_out.as<double>();
        HIP_OK(hipMemcpyAsync(dK, &hc, sizeof(OwConsts), hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(d_jobs, hj.data(), sizeof(owdev::OwJobDev) * n_jobs, hipMemcpyHostToDevice, st));
        owdev::k_note_table<<<dim3(1), dim3(64), 0, st>>>(d_nt);
        const JobChainCfg cc{cfg->sample_rate, cfg->device, cfg->preamp_kind, cfg->power_amp_kind, cfg->no_rail_sag};
        // Voices and chain side by side when the chain is the plain legacy one and leaves room on the chip: a job's run time is serial
        // latency in both kernels, so the 13 % the voices take are hidden behind the chain instead of in front of it.
        const bool overlap = job_chain_is_plain_legacy(cc, hj) && job_voice_overlap(n_jobs);
        DevMem m_prog;
        StreamOwner so2;
        hipEvent_t ev_ready = nullptr, ev_voice = nullptr;
        struct EvGuard { hipEvent_t* a; hipEvent_t* b; ~EvGuard() { if (*a) hipEventDestroy(*a); if (*b) hipEventDestroy(*b); } } evg{&ev_ready, &ev_voice};
        int* d_prog = nullptr;
        if (overlap) {
            m_prog.alloc(sizeof(int) * (vblocks + 1));                 // progress of every voice block + the chain's "gave up waiting" flag
            d_prog = m_prog.as<int>();
            HIP_OK(hipMemsetAsync(d_prog, 0, sizeof(int) * (vblocks + 1), st));
            HIP_OK(hipStreamCreateWithFlags(&so2.s, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&ev_voice, hipEventDisableTiming));
            HIP_OK(hipEventRecord(ev_ready, st));
            HIP_OK(hipStreamWaitEvent(so2.s, ev_ready, 0));
            owdev::k_job_voice<<<dim3((unsigned)vblocks), dim3(64), 0, so2.s>>>(dK, d_nt, d_vrec, d_jobs, d_reed, (int)n_jobs, (long long)n, (long long)stride, d_prog);
            HIP_OK(hipGetLastError());
            HIP_OK(hipEventRecord(ev_voice, so2.s));
        } else {
            owdev::k_job_voice<<<dim3((unsigned)vblocks), dim3(64), 0, st>>>(dK, d_nt, d_vrec, d_jobs, d_reed, (int)n_jobs, (long long)n, (long long)stride);
            HIP_OK(hipGetLastError());
        }
        run_job_chain(cc, dK, hj, d_jobs, d_reed, d_out, n_jobs, (long long)n, (long long)stride, st, d_prog);
        if (overlap) {
            HIP_OK(hipStreamWaitEvent(st, ev_voice, 0));
            int gave_up = 0;
            HIP_OK(hipMemcpyAsync(&gave_up, d_prog + vblocks, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            if (gave_up) run_job_chain(cc, dK, hj, d_jobs, d_reed, d_out, n_jobs, (long long)n, (long long)stride, st);   // the voices are complete now
        }
        if (!out_is_device) HIP_OK(hipMemcpyAsync(out, d_out, sizeof(double) * n_jobs * stride, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        return (long long)n;
    } catch (const std::exception& ex) { set_err(std::string("ow_batch_render: ") + ex.what()); return -1; }
}

void* ow_device_alloc(size_t bytes, int device) {
    void* ptr = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&ptr, bytes ? bytes : 1) != hipSuccess) { set_err("ow_device_alloc: hipMalloc failed"); return nullptr; }
    return ptr;
}
void ow_device_free(void* ptr, int device) {
    if (ptr && hipSetDevice(device) == hipSuccess) hipFree(ptr);
}

void* ow_host_alloc(size_t bytes, int device) {
    void* ptr = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipHostMalloc(&ptr, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { set_err("ow_host_alloc: hipHostMalloc failed"); return nullptr; }
    void* dptr = nullptr;
    if (hipHostGetDevicePointer(&dptr, ptr, 0) != hipSuccess) dptr = nullptr;
    try { host_block_register(ptr, bytes ? bytes : 1, dptr); } catch (...) {}
    return ptr;
}
void ow_host_free(void* ptr, int device) {
    if (!ptr) return;
    host_block_forget(ptr);
    if (hipSetDevice(device) == hipSuccess) hipHostFree(ptr);
}

// ---- ML-pipeline stage after the batch render ---------------------------------------------------------------
int ow_wav24_quantize(const double* samples, size_t n, double scale, int mode, int32_t* out) {
    if ((!samples || !out) && n) { set_err("ow_wav24_quantize: null argument"); return -1; }
    if (mode != OW_WAV_ROUND && mode != OW_WAV_TRUNCATE) { set_err("ow_wav24_quantize: unknown mode"); return -1; }
    const double mx = 8388607.0;
    for (size_t i = 0; i < n; ++i) {
        double v;
        if (mode == OW_WAV_ROUND) {          // main.rs:951-954: round half away from zero, saturating cast, clamp
            v = std::round(samples[i] * scale * mx);
        } else {                             // reed-renderer main.rs:119-123: clamp to +-1, scale, truncate toward zero
            const double c = samples[i] < -1.0 ? -1.0 : (samples[i] > 1.0 ? 1.0 : samples[i]);
            v = std::trunc(c * mx);
        }
        if (!(v == v)) v = 0.0;              // Rust `as i32`: NaN -> 0
        out[i] = (int32_t)(v < -mx ? -mx : (v > mx ? mx : v));
    }
    return 0;
}

int ow_wav24_write(const char* path, const double* samples, size_t n, uint32_t sample_rate, double scale, int mode) {
    if (!path || (!samples && n)) { set_err("ow_wav24_write: null argument"); return -1; }
    std::vector<int32_t> q(n);
    if (ow_wav24_quantize(samples, n, scale, mode, q.data()) != 0) return -1;
    const uint32_t data_bytes = (uint32_t)(n * 3);
    std::vector<uint8_t> buf;
    buf.reserve(68 + data_bytes + 1);
    auto u16 = [&](uint32_t v) { buf.push_back((uint8_t)v); buf.push_back((uint8_t)(v >> 8)); };
    auto u32 = [&](uint32_t v) { u16(v & 0xFFFFu); u16(v >> 16); };
    auto tag = [&](const char* t) { buf.insert(buf.end(), t, t + 4); };
    tag("RIFF"); u32(4 + (8 + 40) + (8 + data_bytes + (data_bytes & 1))); tag("WAVE");
    tag("fmt "); u32(40);
    u16(0xFFFE);                 // WAVE_FORMAT_EXTENSIBLE
    u16(1);                      // channels
    u32(sample_rate);
    u32(sample_rate * 3);        // bytes per second
    u16(3);                      // block align
    u16(24);                     // bits per sample (container)
    u16(22);                     // cbSize
    u16(24);                     // valid bits
    u32(0x4);                    // channel mask: front centre
    static const uint8_t pcm_guid[16] = {0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71};
    buf.insert(buf.end(), pcm_guid, pcm_guid + 16);
    tag("data"); u32(data_bytes);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t v = (uint32_t)q[i];
        buf.push_back((uint8_t)v); buf.push_back((uint8_t)(v >> 8)); buf.push_back((uint8_t)(v >> 16));
    }
    if (data_bytes & 1) buf.push_back(0);
    FILE* f = std::fopen(path, "wb");
    if (!f) { set_err(std::string("ow_wav24_write: cannot open ") + path); return -2; }
    const bool ok = std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    if (std::fclose(f) != 0 || !ok) { set_err(std::string("ow_wav24_write: short write to ") + path); return -3; }
    return 0;
}

int ow_extract_harmonics(const double* audio, size_t n_rows, size_t stride, double sample_rate, const ow_segment* segs, size_t n_segs,
                         double search_pct, int wav24_mode, int device, int audio_is_device, double* amps, double* freqs, double* rms) {
    try {
        if (!audio || !segs || !amps || !freqs) throw std::runtime_error("null argument");
        if (wav24_mode != OW_WAV_NONE && wav24_mode != OW_WAV_ROUND && wav24_mode != OW_WAV_TRUNCATE) throw std::runtime_error("unknown wav24_mode");
        if (!(sample_rate > 0.0) || !(search_pct >= 0.0)) throw std::runtime_error("invalid sample rate or search band");
        if (n_segs == 0) return 0;
        std::vector<owdev::OwSegDev> hs(n_segs);
        std::vector<owdev::OwBinsDev> hb;
        std::vector<double> vals(n_segs);
        uint64_t off = 0;
        for (size_t i = 0; i < n_segs; ++i) {
            const ow_segment& g = segs[i];
            if (g.row >= n_rows || g.end <= g.start || g.end > stride || g.n_harmonics > OW_MAX_HARMONICS)
                throw std::runtime_error("segment " + std::to_string(i) + " out of range");
            const uint32_t n = g.end - g.start;
            if (n > (1u << 22)) throw std::runtime_error("segment longer than 2^22 samples");
            hs[i].row = g.row; hs[i].start = g.start; hs[i].n = n; hs[i].n_harm = g.n_harmonics; hs[i].xw_off = off;
            off += n;
            // numpy's axis: rfftfreq(nfft, d = 1/sr) = arange(nfft/2 + 1) * (1 / (nfft * d)); mask = (axis >= f_lo) & (axis <= f_hi)
            const uint64_t nfft = 4ull * n, nb = nfft / 2 + 1;
            const double val = 1.0 / ((double)nfft * (1.0 / sample_rate));
            vals[i] = val;
            for (uint32_t h = 0; h < g.n_harmonics; ++h) {
                owdev::OwBinsDev b;
                b.seg = (uint32_t)i; b.k_lo = 1; b.k_hi = 0; b.pad = 0;
                const double fh = g.f0 * (double)(h + 1);
                if (!(fh >= sample_rate / 2 - 100)) {
                    const double f_lo = fh * (1.0 - search_pct), f_hi = fh * (1.0 + search_pct);
                    double q = f_lo / val;
                    uint64_t k = (q > 0.0 && q < (double)nb) ? (uint64_t)q : (q >= (double)nb ? nb : 0);
                    while (k > 0 && (double)(k - 1) * val >= f_lo) --k;
                    while (k < nb && (double)k * val < f_lo) ++k;
                    const uint64_t k_lo = k;
                    while (k < nb && (double)k * val <= f_hi) ++k;
                    if (k > k_lo) { b.k_lo = (uint32_t)k_lo; b.k_hi = (uint32_t)(k - 1); }
                }
                hb.push_back(b);
            }
        }
        HIP_OK(hipSetDevice(device));
        StreamOwner so;
        HIP_OK(hipStreamCreateWithFlags(&so.s, hipStreamNonBlocking));
        hipStream_t st = so.s;
        DevMem own_audio, m_xw, m_ss, m_segs, m_bins, m_peaks;   // released on every exit path
        const double* d_audio = audio;
        if (!audio_is_device) {
            own_audio.alloc(sizeof(double) * n_rows * stride);
            HIP_OK(hipMemcpyAsync(own_audio.p, audio, sizeof(double) * n_rows * stride, hipMemcpyHostToDevice, st));
            d_audio = own_audio.as<double>();
        }
        m_xw.alloc(sizeof(double) * std::max<uint64_t>(off, 1));
        m_ss.alloc(sizeof(double) * n_segs);
        m_segs.alloc(sizeof(owdev::OwSegDev) * n_segs);
        double* d_xw = m_xw.as<double>(); double* d_ss = m_ss.as<double>();
        owdev::OwSegDev* d_segs = m_segs.as<owdev::OwSegDev>();
        HIP_OK(hipMemcpyAsync(d_segs, hs.data(), sizeof(owdev::OwSegDev) * n_segs, hipMemcpyHostToDevice, st));
        owdev::k_feat_window<<<dim3((unsigned)n_segs), dim3(256), 0, st>>>(d_audio, stride, d_segs, d_xw, d_ss, wav24_mode);
        std::vector<owdev::OwPeakDev> peaks(hb.size());
        if (!hb.empty()) {
            m_bins.alloc(sizeof(owdev::OwBinsDev) * hb.size());
            m_peaks.alloc(sizeof(owdev::OwPeakDev) * hb.size());
            HIP_OK(hipMemcpyAsync(m_bins.p, hb.data(), sizeof(owdev::OwBinsDev) * hb.size(), hipMemcpyHostToDevice, st));
            owdev::k_feat_peaks<<<dim3((unsigned)hb.size()), dim3(256), 0, st>>>(d_segs, m_bins.as<owdev::OwBinsDev>(), d_xw, m_peaks.as<owdev::OwPeakDev>());
            HIP_OK(hipMemcpyAsync(peaks.data(), m_peaks.p, sizeof(owdev::OwPeakDev) * hb.size(), hipMemcpyDeviceToHost, st));
        }
        std::vector<double> ss(n_segs);
        HIP_OK(hipMemcpyAsync(ss.data(), d_ss, sizeof(double) * n_segs, hipMemcpyDeviceToHost, st));
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(st));
        size_t bi = 0;
        for (size_t i = 0; i < n_segs; ++i) {
            const double N = (double)hs[i].n;
            for (uint32_t h = 0; h < OW_MAX_HARMONICS; ++h) {
                double a = 0.0, f = 0.0;
                if (h < hs[i].n_harm) {
                    const owdev::OwBinsDev& b = hb[bi];
                    if (b.k_lo <= b.k_hi) {
                        a = std::hypot(peaks[bi].re, peaks[bi].im) * 2.0 / N / 0.5;   // np.abs(rfft) * 2.0 / N / 0.5
                        f = (double)peaks[bi].k * vals[i];
                    } else { a = 1e-20; f = segs[i].f0 * (double)(h + 1); }
                    ++bi;
                }
                amps[i * OW_MAX_HARMONICS + h] = a; freqs[i * OW_MAX_HARMONICS + h] = f;
            }
            if (rms) rms[i] = std::max(std::sqrt(ss[i] / N), 1e-20);
        }
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_extract_harmonics: ") + ex.what()); return -1; }
}

}  // extern "C"

// ---- click-band alias audit (alias_audit.rs) -----------------------------------------------------
namespace {
constexpr double AUDIT_SR = 44100.0, AUDIT_RENDER_S = 1.5, AUDIT_ANALYZE_S = 0.5;   // alias_audit.rs:47-53
constexpr uint32_t AUDIT_PROBES = 112;   // nominal + the 0.1 Hz walk over +-5 Hz (101 or 102 points), rounded up
inline double audit_db(double mag) { return mag > 0.0 ? 20.0 * std::log10(mag) : -200.0; }   // mag_to_db :242-248
owdev::OwAuditBq audit_biquad(bool highpass, double fc, double q, double sr) {   // filters.rs:24-38 (RBJ), host libm like the reference
    const double w0 = 2.0 * 3.14159265358979323846 * fc / sr;
    const double cw = std::cos(w0), sw = std::sin(w0);
    const double alpha = sw / (2.0 * q), a0 = 1.0 + alpha;
    owdev::OwAuditBq c;
    if (highpass) { c.b0 = ((1.0 + cw) / 2.0) / a0; c.b1 = (-(1.0 + cw)) / a0; c.b2 = ((1.0 + cw) / 2.0) / a0; }
    else          { c.b0 = ((1.0 - cw) / 2.0) / a0; c.b1 = (1.0 - cw) / a0;    c.b2 = ((1.0 - cw) / 2.0) / a0; }
    c.a1 = (-2.0 * cw) / a0;
    c.a2 = (1.0 - alpha) / a0;
    return c;
}

void alias_audit_analyze_device(const double* d_sig, size_t n_sig, size_t stride, size_t len, double sr, const double* nominal_f0,
                                hipStream_t st, ow_alias_audit_result* out) {
    const size_t analyze_n = (size_t)(sr * AUDIT_ANALYZE_S);   // alias_audit.rs:166
    if (len < analyze_n)
        throw std::runtime_error("alias_audit signal too short: " + std::to_string(len) + " samples for " + std::to_string(analyze_n) + " analysis window");
    if (analyze_n == 0 || analyze_n > 0xffffffffull || n_sig > 65535) throw std::runtime_error("analysis window or signal count out of range");
    const size_t tail_off = len - analyze_n;
    const double nn = (double)analyze_n;
    const double two_pi = 2.0 * 3.14159265358979323846;
    auto magnitude = [&](double2 v) { const double a = v.x / nn, b = v.y / nn; return 2.0 * std::sqrt(a * a + b * b); };   // :239

    // refine_f0 (:252-265): the probe list is the reference's own loop (nominal first, then f += 0.1 while f <= nominal + 5)
    std::vector<double> freq(n_sig * AUDIT_PROBES, 0.0), omega(n_sig * AUDIT_PROBES, 0.0);
    std::vector<uint32_t> count(n_sig, 0);
    for (size_t s = 0; s < n_sig; ++s) {
        const double nominal = nominal_f0[s];
        uint32_t k = 0;
        freq[s * AUDIT_PROBES + k++] = nominal;
        double f = nominal - 5.0;
        while (f <= nominal + 5.0) {
            if (k >= AUDIT_PROBES) throw std::runtime_error("refine_f0 grid larger than expected");
            freq[s * AUDIT_PROBES + k++] = f;
            f += 0.1;
        }
        count[s] = k;
        for (uint32_t j = 0; j < k; ++j) omega[s * AUDIT_PROBES + j] = two_pi * freq[s * AUDIT_PROBES + j] / sr;   // :233
    }
    DevMem d_omega, d_count, d_out, d_ss;
    d_omega.alloc(sizeof(double) * omega.size());
    d_count.alloc(sizeof(uint32_t) * n_sig);
    d_out.alloc(sizeof(double2) * omega.size());
    d_ss.alloc(sizeof(double) * n_sig);
    HIP_OK(hipMemcpyAsync(d_omega.p, omega.data(), sizeof(double) * omega.size(), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_count.p, count.data(), sizeof(uint32_t) * n_sig, hipMemcpyHostToDevice, st));
    owdev::k_audit_dft<<<dim3(AUDIT_PROBES, (unsigned)n_sig), dim3(256), 0, st>>>(d_sig, stride, tail_off, (uint32_t)analyze_n, d_omega.as<double>(),
                                                                                 d_count.as<uint32_t>(), AUDIT_PROBES, d_out.as<double2>());
    // bandpass_rms (:270-282) has no dependence on f0: same stream, behind the first DFT pass
    owdev::OwAuditBand band;
    band.hp = audit_biquad(true, 5000.0, 0.70710678118654752440, sr);     // HF_BAND_LO_HZ :62
    band.lp = audit_biquad(false, 18000.0, 0.70710678118654752440, sr);   // HF_BAND_HI_HZ :64
    owdev::k_audit_bandpass<<<dim3((unsigned)((n_sig + 63) / 64)), dim3(64), 0, st>>>(d_sig, stride, tail_off, (uint32_t)analyze_n, (uint32_t)n_sig, band,
                                                                                      d_ss.as<double>());
    std::vector<double2> spec(omega.size());
    HIP_OK(hipMemcpyAsync(spec.data(), d_out.p, sizeof(double2) * spec.size(), hipMemcpyDeviceToHost, st));
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(st));
    std::vector<double> f0(n_sig);
    for (size_t s = 0; s < n_sig; ++s) {
        double best_f = freq[s * AUDIT_PROBES], best = magnitude(spec[s * AUDIT_PROBES]);
        for (uint32_t j = 1; j < count[s]; ++j) {
            const double m = magnitude(spec[s * AUDIT_PROBES + j]);
            if (m > best) { best = m; best_f = freq[s * AUDIT_PROBES + j]; }
        }
        f0[s] = best_f;
        for (uint32_t k = 0; k < OW_AUDIT_HARMONICS; ++k) omega[s * AUDIT_PROBES + k] = two_pi * ((double)(k + 1) * best_f) / sr;   // :180
        count[s] = OW_AUDIT_HARMONICS;
    }
    HIP_OK(hipMemcpyAsync(d_omega.p, omega.data(), sizeof(double) * omega.size(), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_count.p, count.data(), sizeof(uint32_t) * n_sig, hipMemcpyHostToDevice, st));
    owdev::k_audit_dft<<<dim3(OW_AUDIT_HARMONICS, (unsigned)n_sig), dim3(256), 0, st>>>(d_sig, stride, tail_off, (uint32_t)analyze_n, d_omega.as<double>(),
                                                                                       d_count.as<uint32_t>(), AUDIT_PROBES, d_out.as<double2>());
    std::vector<double> ss(n_sig);
    HIP_OK(hipMemcpyAsync(spec.data(), d_out.p, sizeof(double2) * spec.size(), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(ss.data(), d_ss.p, sizeof(double) * n_sig, hipMemcpyDeviceToHost, st));
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(st));
    for (size_t s = 0; s < n_sig; ++s) {
        ow_alias_audit_result& r = out[s];
        std::memset(&r, 0, sizeof r);
        const double h1 = magnitude(spec[s * AUDIT_PROBES]);   // dft_magnitude(tail, f0) :178 == the k = 0 probe
        for (uint32_t k = 0; k < OW_AUDIT_HARMONICS; ++k) {
            const double mag = magnitude(spec[s * AUDIT_PROBES + k]);
            r.harmonic_db[k] = audit_db(mag);
            r.harmonic_dbc[k] = h1 > 0.0 ? 20.0 * std::log10(mag / h1) : -200.0;
        }
        r.harmonic_dbc[0] = 0.0;
        double worst = -INFINITY;
        uint32_t worst_from = 6;                              // plateau_metric :213-227, PLATEAU_FIRST..LAST_HARMONIC = 6..11
        for (uint32_t i = 5; i < 10; ++i) {
            const double delta = r.harmonic_dbc[i + 1] - r.harmonic_dbc[i];
            if (delta > worst) { worst = delta; worst_from = i + 1; }
        }
        r.f0_hz = f0[s];
        r.h1_dbfs = audit_db(h1);
        r.max_step_up_db = worst;
        r.max_step_up_from_harmonic = worst_from;
        const double hf_rms = std::sqrt(ss[s] / nn);
        r.hf_band_dbc = h1 > 0.0 ? 20.0 * std::log10(hf_rms / h1) : -200.0;
    }
}
}  // namespace

extern "C" {

int ow_alias_audit_analyze(const double* signals, size_t n_signals, size_t stride, size_t len, double sample_rate,
                           const double* nominal_f0, int device, int signals_is_device, ow_alias_audit_result* out) {
    try {
        if (!signals || !nominal_f0 || !out) throw std::runtime_error("null argument");
        if (!(sample_rate > 0.0) || len > stride) throw std::runtime_error("invalid sample rate or row length");
        if (n_signals == 0) return 0;
        HIP_OK(hipSetDevice(device));
        StreamOwner so;
        HIP_OK(hipStreamCreateWithFlags(&so.s, hipStreamNonBlocking));
        DevMem d_sig;
        const double* src = signals;
        if (!signals_is_device) {
            d_sig.alloc(sizeof(double) * n_signals * stride);
            HIP_OK(hipMemcpyAsync(d_sig.p, signals, sizeof(double) * n_signals * stride, hipMemcpyHostToDevice, so.s));
            src = d_sig.as<double>();
        }
        alias_audit_analyze_device(src, n_signals, stride, len, sample_rate, nominal_f0, so.s, out);
        return 0;
    } catch (const std::exception& ex) { set_err(std::string("ow_alias_audit_analyze: ") + ex.what()); return -1; }
}

int ow_alias_audit_run(const uint8_t* notes, const uint8_t* velocities, size_t n, int device, int preamp_kind,
                       ow_alias_audit_result* out, double* signals_out, size_t signals_stride) {
    ow_pool* pool = nullptr;
    try {
        if (!notes || !velocities || !out) throw std::runtime_error("null argument");
        if (n == 0) return 0;
        const size_t total = (size_t)(AUDIT_SR * AUDIT_RENDER_S);   // :151
        if (signals_out && signals_stride < total) throw std::runtime_error("signals_stride shorter than the render");
        g_err.clear();
        pool = ow_pool_new(AUDIT_SR, n, device, preamp_kind);       // WurliEngine::new(sr) per stimulus, :137
        if (!pool) throw std::runtime_error(g_err.empty() ? "pool creation failed" : g_err);
        ow_pool_ensure_buffer_capacity(pool, 1024);                 // :138
        for (size_t k = 0; k < n; ++k) {                            // :139-143
            ow_engine* e = pool->engines[k];
            ow_engine_set_volume(e, 0.5);
            ow_engine_set_tremolo_depth(e, 0.0);
            ow_engine_set_speaker_character(e, 0.0);
            ow_engine_set_mlp_enabled(e, 1);
            ow_engine_set_noise_enabled(e, 0);
        }
        for (int k = 0; k < 6; ++k) ow_pool_render(pool, nullptr, 0, 1024);   // smoother settle, :147-150
        std::vector<double> nominal(n);
        for (size_t k = 0; k < n; ++k) {
            ow_engine_note_on(pool->engines[k], notes[k], (float)velocities[k] / 127.0f);   // :152
            nominal[k] = 440.0 * std::pow(2.0, ((double)notes[k] - 69.0) / 12.0);            // midi_note_hz :284-287
        }
        DevMem d_sig;
        d_sig.alloc(sizeof(double) * n * total);
        size_t pos = 0;
        while (pos < total) {                                       // :154-165
            const size_t len = std::min<size_t>(1024, total - pos);
            ow_pool_render(pool, nullptr, 0, len);
            owdev::k_audit_gather<<<dim3((unsigned)((len + 255) / 256), (unsigned)n), dim3(256), 0, pool->stream>>>(
                pool->d_out, pool->out_ld, d_sig.as<double>(), total, pos, (uint32_t)len);
            pos += len;
        }
        HIP_OK(hipGetLastError());
        if (!g_err.empty()) throw std::runtime_error(g_err);
        if (signals_out)
            HIP_OK(hipMemcpy2DAsync(signals_out, signals_stride * sizeof(double), d_sig.p, total * sizeof(double), total * sizeof(double), n,
                                    hipMemcpyDeviceToHost, pool->stream));
        alias_audit_analyze_device(d_sig.as<double>(), n, total, total, AUDIT_SR, nominal.data(), pool->stream, out);
        pool_destroy(pool);
        return 0;
    } catch (const std::exception& ex) {
        if (pool) pool_destroy(pool);
        set_err(std::string("ow_alias_audit_run: ") + ex.what());
        return -1;
    }
}

}  // extern "C"

// ---- `preamp-bench render-midi` ---------------------------------------------------------------------
namespace {
struct SmfReader {   // the subset of the SMF grammar cmd_render_midi consumes through midly (main.rs:1627-1708)
    const uint8_t* d; size_t n;
    uint32_t be32(size_t p) const { return ((uint32_t)d[p] << 24) | ((uint32_t)d[p + 1] << 16) | ((uint32_t)d[p + 2] << 8) | (uint32_t)d[p + 3]; }
    static uint32_t vlq(const uint8_t* d, size_t& p, size_t end) {
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            if (p >= end) throw std::runtime_error("truncated track");
            const uint8_t b = d[p++];
            v = (v << 7) | (uint32_t)(b & 0x7F);
            if (!(b & 0x80)) return v;
        }
        throw std::runtime_error("variable-length quantity longer than four bytes");
    }
    std::vector<ow_timed_event> events(int track_filter) const {
        if (n < 14 || std::memcmp(d, "MThd", 4) != 0) throw std::runtime_error("not a Standard MIDI File");
        const uint32_t hlen = be32(4);
        if (hlen < 6 || 8 + (size_t)hlen > n) throw std::runtime_error("bad MThd chunk");
        const uint32_t division = ((uint32_t)d[12] << 8) | d[13];
        if (division & 0x8000u) throw std::runtime_error("Only metrical (ticks per beat) MIDI timing is supported");   // main.rs:1630-1636
        const double ticks_per_beat = (double)division;
        std::vector<ow_timed_event> out;
        int track_idx = 0;
        for (size_t pos = 8 + hlen; pos + 8 <= n;) {
            const bool is_track = std::memcmp(d + pos, "MTrk", 4) == 0;
            const size_t len = be32(pos + 4);
            pos += 8;
            if (pos + len > n) throw std::runtime_error("truncated chunk");
            if (is_track) {
                track(pos, pos + len, ticks_per_beat, track_filter < 0 || track_filter == track_idx, out);
                ++track_idx;
            }
            pos += len;
        }
        return out;
    }
    void track(size_t p, size_t end, double ticks_per_beat, bool emit_notes, std::vector<ow_timed_event>& out) const {
        double tempo = 500000.0, time_s = 0.0;   // per track: 120 BPM until this track's own tempo events (main.rs:1653-1654)
        uint8_t running = 0;
        while (p < end) {
            const uint64_t delta_ticks = vlq(d, p, end);
            time_s += ((double)delta_ticks / ticks_per_beat) * (tempo / 1000000.0);   // main.rs:1661-1663
            if (p >= end) throw std::runtime_error("truncated track");
            uint8_t status = d[p];
            if (status & 0x80) ++p;
            else if (running) status = running;
            else throw std::runtime_error("data byte without running status");
            if (status == 0xFF) {
                if (p >= end) throw std::runtime_error("truncated track");
                const uint8_t type = d[p++];
                const uint32_t l = vlq(d, p, end);
                if (p + l > end) throw std::runtime_error("truncated track");
                if (type == 0x51 && l == 3) tempo = (double)(((uint32_t)d[p] << 16) | ((uint32_t)d[p + 1] << 8) | (uint32_t)d[p + 2]);
                p += l;
                running = 0;
            } else if (status == 0xF0 || status == 0xF7) {
                const uint32_t l = vlq(d, p, end);
                if (p + l > end) throw std::runtime_error("truncated track");
                p += l;
                running = 0;
            } else if (status >= 0xF0) {
                throw std::runtime_error("system message inside a track");
            } else {
                running = status;
                const uint8_t kind = status & 0xF0;
                const size_t nd = (kind == 0xC0 || kind == 0xD0) ? 1 : 2;
                if (p + nd > end) throw std::runtime_error("truncated track");
                const uint8_t a = d[p] & 0x7F, b = nd == 2 ? (uint8_t)(d[p + 1] & 0x7F) : (uint8_t)0;
                p += nd;
                if (!emit_notes) continue;
                ow_timed_event e{};
                e.time_s = time_s;
                if (kind == 0x90)      { e.type = b == 0 ? 1 : 0; e.note = a; e.value = b; }       // velocity 0 = note-off (main.rs:1671-1678)
                else if (kind == 0x80) { e.type = 1; e.note = a; }
                else if (kind == 0xB0 && a == 64) { e.type = 2; e.value = b >= 64 ? 1 : 0; }      // sustain pedal (main.rs:1693-1703)
                else continue;
                out.push_back(e);
            }
        }
    }
};
}  // namespace

extern "C" {

long long ow_smf_parse(const uint8_t* data, size_t len, int track_filter, ow_timed_event* out, size_t cap) {
    try {
        if (!data || (!out && cap)) throw std::runtime_error("null argument");
        const std::vector<ow_timed_event> ev = SmfReader{data, len}.events(track_filter);
        for (size_t i = 0; i < std::min(cap, ev.size()); ++i) out[i] = ev[i];
        return (long long)ev.size();
    } catch (const std::exception& ex) { set_err(std::string("ow_smf_parse: ") + ex.what()); return -1; }
}

long long ow_render_midi(const ow_timed_event* events, const size_t* job_offsets, size_t n_jobs, const ow_midi_render_cfg* cfg,
                         double* out, size_t stride, ow_midi_render_stats* stats) {
    try {
        if (!job_offsets || !cfg || (!out && !stats)) throw std::runtime_error("null argument");
        if (cfg->struct_size != sizeof(ow_midi_render_cfg))
            throw std::runtime_error("ABI mismatch: ow_midi_render_cfg.struct_size does not match this library's openwurli_hip.h (OW_ABI_VERSION " +
                                     std::to_string(OW_ABI_VERSION) + ")");
        if (cfg->preamp_kind != OW_PREAMP_LEGACY8 && cfg->preamp_kind != OW_PREAMP_MELANGE12) throw std::runtime_error("unknown preamp_kind");
        if (n_jobs == 0) return 0;
        const double SR = 44100.0;                       // BASE_SR, main.rs:27
        const size_t n_ev = job_offsets[n_jobs];
        if (n_ev && !events) throw std::runtime_error("null argument");
        std::vector<owdev::OwMidiEvDev> hev(std::max<size_t>(n_ev, 1));
        std::vector<owdev::OwMidiJobDev> hj(n_jobs);
        size_t longest = 0;
        for (size_t j = 0; j < n_jobs; ++j) {
            const size_t b = job_offsets[j], e = job_offsets[j + 1];
            if (e < b || e > n_ev) throw std::runtime_error("job_offsets not monotonic");
            std::vector<ow_timed_event> ev(events + b, events + e);
            for (const ow_timed_event& x : ev)
                if (!(x.time_s == x.time_s) || x.type > 2) throw std::runtime_error("event with NaN time or unknown type");   // partial_cmp().unwrap() panics
            std::stable_sort(ev.begin(), ev.end(), [](const ow_timed_event& a, const ow_timed_event& c) { return a.time_s < c.time_s; });   // :1712
            size_t total = 0;
            if (!ev.empty()) {
                const double x = (ev.back().time_s + cfg->tail_s) * SR;                                                     // :1719-1721
                total = (!(x == x) || x <= 0.0) ? 0 : (x >= 1.8446744073709552e19 ? SIZE_MAX : (size_t)x);
                if (total > (size_t)1 << 36) throw std::runtime_error("render longer than 2^36 samples");
            }
            for (size_t i = 0; i < ev.size(); ++i) {
                // the chunk at sample_pos fires every event with time_s <= sample_pos / SR (:1778-1781)
                const double t = ev[i].time_s;
                uint64_t c = t > 0.0 ? (uint64_t)(t * SR / 64.0) : 0;
                while (c > 0 && (double)(64 * (c - 1)) / SR >= t) --c;
                while ((double)(64 * c) / SR < t) ++c;
                owdev::OwMidiEvDev& dv = hev[b + i];
                dv.chunk = (uint32_t)std::min<uint64_t>(c, 0xFFFFFFFFull);
                dv.type = ev[i].type; dv.note = ev[i].note; dv.value = ev[i].value; dv.pad = 0;
            }
            hj[j].ev_begin = b; hj[j].n_events = (uint32_t)ev.size(); hj[j].pad = 0; hj[j].total_samples = total;
            if (stats) { stats[j].n_samples = total; stats[j].note_ons = 0; stats[j].peak_polyphony = 0; }
            longest = std::max(longest, total);
        }
        if (!out || longest == 0) return (long long)longest;
        if (stride < longest) throw std::runtime_error("stride smaller than the longest job");
        int ndev = 0;
        HIP_OK(hipGetDeviceCount(&ndev));
        if (ndev <= 0) throw std::runtime_error("no HIP device: openwurli-hip has no CPU fallback");
        HIP_OK(hipSetDevice(cfg->device));
        OwConsts hc;
        owhip::build_consts(hc, SR, cfg->preamp_kind);
        std::vector<owdev::OwJobDev> hjob(n_jobs);       // chain parameters: Speaker(speaker), static 1 Mohm LDR, volume, power amp
        for (auto& q : hjob) {
            std::memset(&q, 0, sizeof q);
            q.note = 60; q.mlp = 1; q.poweramp = cfg->no_poweramp ? 0 : 1; q.volume = cfg->volume; q.speaker = cfg->speaker;
            // main.rs:1752-1754 sets 1 Mohm and THEN calls reset(): the legacy solver keeps its resistance, the melange adapter's reset() returns
            // to the settled clone at the nominal pot (melange_adapter.rs:88-93) and the command never sets it again
            q.r_ldr = cfg->preamp_kind == OW_PREAMP_MELANGE12 ? 9.99999999999999854e4 : 1000000.0;
        }
        StreamOwner so;
        HIP_OK(hipStreamCreateWithFlags(&so.s, hipStreamNonBlocking));
        hipStream_t st = so.s;
        DevMem dK, d_nt, d_vrec, d_jobs, d_ev, d_held, d_sum, d_out, d_chain, d_stats;
        dK.alloc(sizeof(OwConsts));
        d_nt.alloc(sizeof(double) * NT_COUNT * 64);
        d_vrec.alloc(sizeof(double) * n_jobs * OW_VREC_DOUBLES);
        d_jobs.alloc(sizeof(owdev::OwMidiJobDev) * n_jobs);
        d_ev.alloc(sizeof(owdev::OwMidiEvDev) * hev.size());
        d_held.alloc(sizeof(uint32_t) * hev.size());
        d_sum.alloc(sizeof(double) * n_jobs * longest);
        d_out.alloc(sizeof(double) * n_jobs * longest);
        d_chain.alloc(sizeof(owdev::OwJobDev) * n_jobs);
        d_stats.alloc(sizeof(owdev::OwMidiStatsDev) * n_jobs);
        HIP_OK(hipMemcpyAsync(dK.p, &hc, sizeof(OwConsts), hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(d_jobs.p, hj.data(), sizeof(owdev::OwMidiJobDev) * n_jobs, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(d_ev.p, hev.data(), sizeof(owdev::OwMidiEvDev) * hev.size(), hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(d_chain.p, hjob.data(), sizeof(owdev::OwJobDev) * n_jobs, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(d_sum.p, 0, sizeof(double) * n_jobs * longest, st));   // rows behind a shorter job's end feed the chain zeros
        owdev::k_note_table<<<dim3(1), dim3(64), 0, st>>>(d_nt.as<double>());
        owdev::k_midi_voices<<<dim3((unsigned)n_jobs), dim3(64), 0, st>>>(dK.as<OwConsts>(), d_nt.as<double>(), d_vrec.as<double>(), d_jobs.as<owdev::OwMidiJobDev>(),
                                                                         d_ev.as<owdev::OwMidiEvDev>(), d_held.as<uint32_t>(), d_sum.as<double>(), (long long)longest,
                                                                         d_stats.as<owdev::OwMidiStatsDev>());
        HIP_OK(hipGetLastError());
        const JobChainCfg cc{SR, cfg->device, cfg->preamp_kind, cfg->power_amp_kind, cfg->no_rail_sag};
        run_job_chain(cc, dK.as<OwConsts>(), hjob, d_chain.as<owdev::OwJobDev>(), d_sum.as<double>(), d_out.as<double>(), n_jobs, (long long)longest, (long long)longest, st);
        std::vector<owdev::OwMidiStatsDev> hs(n_jobs);
        HIP_OK(hipMemcpy2DAsync(out, stride * sizeof(double), d_out.p, longest * sizeof(double), longest * sizeof(double), n_jobs, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(hs.data(), d_stats.p, sizeof(owdev::OwMidiStatsDev) * n_jobs, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        for (size_t j = 0; j < n_jobs; ++j) {
            // the chain kernel ran every row to the longest job's length: what lies behind a job's own end is not part of its render
            for (size_t i = hj[j].total_samples; i < std::min(stride, longest); ++i) out[j * stride + i] = 0.0;
            if (stats) { stats[j].note_ons = hs[j].note_ons; stats[j].peak_polyphony = hs[j].peak_polyphony; }
        }
        return (long long)longest;
    } catch (const std::exception& ex) { set_err(std::string("ow_render_midi: ") + ex.what()); return -1; }
}

}  // extern "C"
