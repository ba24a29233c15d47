/* A plain C99 host of the openwurli-hip C-ABI that plays the part of the reference's nih-plug shell
 * (/root/reference/crates/openwurli-plugin/src/lib.rs): initialize() (:87-100), reset() (:102-104), sync_params() (:36-47),
 * handle_event() (:49-62) and the sample-accurate process() loop (:108-166) -- every call goes through
 * include/openwurli_hip.h, nothing else.  It stands in for the Rust facade that cannot be compiled in this image (no Rust
 * toolchain): what a maintainer's facade would do per audio callback, this program does.
 *
 *   plugin_process render <script> <out.f32> <sample_rate> <max_buffer> <n_buffers>
 *       plays an event script (lines: "<buffer> <timing> <type> <note> <value>"; type 0 NoteOn(note, velocity), 1 NoteOff(note),
 *       2 MidiCC damper pedal(value), 3 param change: note = 0 volume / 1 tremolo depth / 2 speaker character, 4 plugin reset(),
 *       5 = a parameter value the host restored BEFORE initialize(), e.g. from a saved session)
 *       through process() in buffers of <max_buffer> samples and writes channel 0 (== channel 1) as raw little-endian f32.
 *   plugin_process audit <n_engines> <n_renders>
 *       allocation audit of the realtime path (nih-plug assert_process_allocs, plugin/Cargo.toml:21): counts the
 *       heap allocations made BY CODE OF libopenwurli_hip.so (operator new, attributed by return address; the library imports no
 *       malloc-family symbol at all, which the pytest wrapper checks on its dynamic symbol table) and
 *       every hipMalloc / hipFree / hipHostMalloc / hipHostFree / stream / event creation, over <n_renders> ow_pool_render calls
 *       of a pool that is struck, re-struck (whole keyboard) and retargeted inside the window.  Prints the counts; exit 0 iff all 0.
 *   plugin_process fault <n_engines>
 *       fault injection (openwurli_hip_test.h): a failing render must hand back silence in EVERY row of the caller's block,
 *       report through ow_last_error, and the next render must work.
 * Exit code 3 = no usable HIP device (the library has no CPU fallback and says so).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/openwurli_hip.h"
#include "../../include/openwurli_hip_test.h"

/* ------------------------------------------------------------------ allocation audit: interposed allocators */
static volatile int g_window = 0;          /* counting on/off */
static unsigned long g_lib_new = 0;        /* operator new reached from code of libopenwurli_hip.so */
static unsigned long g_hip_alloc = 0;      /* hipMalloc*, hipHostMalloc, hipFree*, hipHostFree, stream / event creation */
static void* g_lib_base = NULL;

static int from_lib(void* ret) {
    Dl_info di;
    return g_lib_base && ret && dladdr(ret, &di) && di.dli_fbase == g_lib_base;
}
#define COUNT_NEW() do { if (g_window && from_lib(__builtin_return_address(0))) ++g_lib_new; } while (0)

/* C++ operator new / delete (Itanium ABI names): the executable's definitions win the dynamic lookup, so std::vector, std::string,
 * std::thread ... inside the library land here.  All of them forward to malloc / free, so pairs stay consistent process-wide. */
void* _Znwm(size_t n) { COUNT_NEW(); void* p = malloc(n ? n : 1); if (!p) abort(); return p; }
void* _Znam(size_t n) { COUNT_NEW(); void* p = malloc(n ? n : 1); if (!p) abort(); return p; }
void* _ZnwmRKSt9nothrow_t(size_t n, const void* t) { (void)t; COUNT_NEW(); return malloc(n ? n : 1); }
void* _ZnamRKSt9nothrow_t(size_t n, const void* t) { (void)t; COUNT_NEW(); return malloc(n ? n : 1); }
void* _ZnwmSt11align_val_t(size_t n, size_t al) { COUNT_NEW(); void* p = NULL; if (posix_memalign(&p, al < sizeof(void*) ? sizeof(void*) : al, n ? n : 1)) abort(); return p; }
void* _ZnamSt11align_val_t(size_t n, size_t al) { COUNT_NEW(); void* p = NULL; if (posix_memalign(&p, al < sizeof(void*) ? sizeof(void*) : al, n ? n : 1)) abort(); return p; }
void _ZdlPv(void* p) { free(p); }
void _ZdaPv(void* p) { free(p); }
void _ZdlPvm(void* p, size_t n) { (void)n; free(p); }
void _ZdaPvm(void* p, size_t n) { (void)n; free(p); }
void _ZdlPvSt11align_val_t(void* p, size_t al) { (void)al; free(p); }
void _ZdaPvSt11align_val_t(void* p, size_t al) { (void)al; free(p); }
void _ZdlPvmSt11align_val_t(void* p, size_t n, size_t al) { (void)n; (void)al; free(p); }
void _ZdlPvRKSt9nothrow_t(void* p, const void* t) { (void)t; free(p); }
void _ZdaPvRKSt9nothrow_t(void* p, const void* t) { (void)t; free(p); }

/* HIP allocation entry points (hipError_t is an int-sized enum; pointers and sizes are plain): counted, then forwarded. */
typedef int (*fn_pp_sz)(void**, size_t);
typedef int (*fn_pp_sz_u)(void**, size_t, unsigned);
typedef int (*fn_p)(void*);
typedef int (*fn_pp)(void**);
typedef int (*fn_pp_u)(void**, unsigned);
#define NEXT(type, name) static type real = NULL; if (!real) *(void**)(&real) = dlsym(RTLD_NEXT, name)   /* POSIX idiom: dlsym -> function pointer */
int hipMalloc(void** p, size_t n) { NEXT(fn_pp_sz, "hipMalloc"); if (g_window) ++g_hip_alloc; return real(p, n); }
int hipHostMalloc(void** p, size_t n, unsigned f) { NEXT(fn_pp_sz_u, "hipHostMalloc"); if (g_window) ++g_hip_alloc; return real(p, n, f); }
int hipFree(void* p) { NEXT(fn_p, "hipFree"); if (g_window) ++g_hip_alloc; return real(p); }
int hipHostFree(void* p) { NEXT(fn_p, "hipHostFree"); if (g_window) ++g_hip_alloc; return real(p); }
int hipStreamCreateWithFlags(void** s, unsigned f) { NEXT(fn_pp_u, "hipStreamCreateWithFlags"); if (g_window) ++g_hip_alloc; return real(s, f); }
int hipEventCreate(void** e) { NEXT(fn_pp, "hipEventCreate"); if (g_window) ++g_hip_alloc; return real(e); }
int hipEventCreateWithFlags(void** e, unsigned f) { NEXT(fn_pp_u, "hipEventCreateWithFlags"); if (g_window) ++g_hip_alloc; return real(e, f); }

/* ------------------------------------------------------------------ the plugin shell, in C */
typedef struct plugin_params {   /* params.rs:5-46 */
    double volume, tremolo_depth, speaker_character, noise_gain;
    int mlp_enabled, noise_enable;
} plugin_params;

typedef struct note_event {      /* the subset of nih-plug's NoteEvent the shell handles (lib.rs:49-62) */
    unsigned timing;             /* sample offset inside the buffer */
    int type;                    /* 0 NoteOn  1 NoteOff  2 MidiCC(damper pedal)  3 parameter automation  4 reset */
    int note;
    float value;
} note_event;

typedef struct plugin {
    plugin_params params;
    ow_engine* engine;
} plugin;

static void sync_params(plugin* pl) {                                   /* lib.rs:36-47 */
    ow_engine_set_volume(pl->engine, pl->params.volume);
    ow_engine_set_tremolo_depth(pl->engine, pl->params.tremolo_depth);
    ow_engine_set_speaker_character(pl->engine, pl->params.speaker_character);
    ow_engine_set_mlp_enabled(pl->engine, pl->params.mlp_enabled);
    ow_engine_set_noise_enabled(pl->engine, pl->params.noise_enable);
    ow_engine_set_noise_gain(pl->engine, pl->params.noise_gain);
}

static void handle_event(plugin* pl, const note_event* ev) {            /* lib.rs:49-62 */
    switch (ev->type) {
        case 0: ow_engine_note_on(pl->engine, (uint8_t)ev->note, ev->value); break;
        case 1: ow_engine_note_off(pl->engine, (uint8_t)ev->note); break;
        case 2: ow_engine_set_sustain(pl->engine, ev->value >= 0.5f); break;
        default: break;
    }
}

static int plugin_new(plugin* pl) {                                     /* Default: WurliEngine::new(44_100.0), lib.rs:23-30 */
    pl->params.volume = 0.5; pl->params.tremolo_depth = 0.5; pl->params.speaker_character = 0.0; pl->params.noise_gain = 1.0;
    pl->params.mlp_enabled = 1; pl->params.noise_enable = 0;
    pl->engine = ow_engine_new(44100.0, 0, OW_PREAMP_LEGACY8);
    return pl->engine != NULL;
}

static void plugin_initialize(plugin* pl, double sample_rate, size_t max_buffer_size) {   /* lib.rs:87-100 */
    ow_engine_set_sample_rate(pl->engine, sample_rate);
    ow_engine_ensure_buffer_capacity(pl->engine, max_buffer_size);
    sync_params(pl);
}

static void plugin_reset(plugin* pl) { ow_engine_reset(pl->engine); }                     /* lib.rs:102-104 */

/* process(), lib.rs:108-166: events sorted by timing; channel 0 is rendered in event-delimited sub-blocks, then fanned out. */
static void plugin_process(plugin* pl, float* const* channels, size_t n_channels, size_t num_samples, const note_event* events, size_t n_events) {
    size_t block_start = 0, next = 0;
    sync_params(pl);
    if (num_samples == 0) return;
    while (block_start < num_samples) {
        size_t block_end, len;
        while (next < n_events && events[next].timing <= block_start) handle_event(pl, &events[next++]);
        block_end = next < n_events ? (events[next].timing < num_samples ? events[next].timing : num_samples) : num_samples;
        len = block_end - block_start;
        if (len > 0) ow_engine_render(pl->engine, channels[0] + block_start, len);
        block_start = block_end;
    }
    while (next < n_events) handle_event(pl, &events[next++]);          /* trailing events */
    for (size_t c = 1; c < n_channels; ++c) memcpy(channels[c], channels[0], num_samples * sizeof(float));
}

/* ------------------------------------------------------------------ mode: render a script */
typedef struct script_line { unsigned buffer; note_event ev; } script_line;

static int cmd_render(const char* script_path, const char* out_path, double sr, size_t max_buffer, size_t n_buffers) {
    FILE* f = fopen(script_path, "r");
    script_line* lines = NULL;
    size_t n_lines = 0, cap = 0, li = 0;
    plugin pl;
    float *left, *right;
    float* chans[2];
    FILE* out;
    if (!f) { fprintf(stderr, "cannot open %s\n", script_path); return 2; }
    for (;;) {
        script_line l;
        int r = fscanf(f, "%u %u %d %d %f", &l.buffer, &l.ev.timing, &l.ev.type, &l.ev.note, &l.ev.value);
        if (r != 5) break;
        if (n_lines == cap) { cap = cap ? 2 * cap : 256; lines = (script_line*)realloc(lines, cap * sizeof *lines); if (!lines) return 2; }
        lines[n_lines++] = l;
    }
    fclose(f);
    if (!plugin_new(&pl)) { fprintf(stderr, "ow_engine_new failed: %s\n", ow_last_error()); return 3; }
    for (size_t i = 0; i < n_lines; ++i)
        if (lines[i].ev.type == 5) {
            if (lines[i].ev.note == 0) pl.params.volume = lines[i].ev.value;
            else if (lines[i].ev.note == 1) pl.params.tremolo_depth = lines[i].ev.value;
            else pl.params.speaker_character = lines[i].ev.value;
        }
    /* nih-plug: initialize(), then reset() before the first process() ("reset() is always called after initialize()"): the setters
     * of initialize()'s sync_params() are followed by reset() with NO render in between -- reset() must snap to those targets */
    plugin_initialize(&pl, sr, max_buffer);
    plugin_reset(&pl);
    left = (float*)calloc(max_buffer, sizeof(float)); right = (float*)calloc(max_buffer, sizeof(float));
    chans[0] = left; chans[1] = right;
    out = fopen(out_path, "wb");
    if (!out || !left || !right) return 2;
    for (size_t b = 0; b < n_buffers; ++b) {
        note_event evs[256];
        size_t n_ev = 0;
        while (li < n_lines && (lines[li].buffer == b || lines[li].ev.type == 5)) {
            const note_event* e = &lines[li].ev;
            if (e->type == 5) {
                /* consumed before initialize() */
            } else if (e->type == 3) {                       /* host automation lands in the param object before process() reads it */
                if (e->note == 0) pl.params.volume = e->value;
                else if (e->note == 1) pl.params.tremolo_depth = e->value;
                else pl.params.speaker_character = e->value;
            } else if (e->type == 4) {
                plugin_reset(&pl);
            } else if (n_ev < 256) {
                evs[n_ev++] = *e;
            }
            ++li;
        }
        plugin_process(&pl, chans, 2, max_buffer, evs, n_ev);
        if (memcmp(left, right, max_buffer * sizeof(float)) != 0) { fprintf(stderr, "channel fan-out mismatch\n"); return 2; }
        fwrite(left, sizeof(float), max_buffer, out);
    }
    fclose(out);
    {
        ow_diag d;
        ow_engine_get_diag(pl.engine, &d);
        printf("{\"buffers\": %zu, \"active_voices\": %u, \"nan_guard_fires\": %llu, \"last_error\": \"%s\"}\n", n_buffers, d.active_voices,
               (unsigned long long)d.nan_guard_fires, ow_last_error());
    }
    ow_engine_free(pl.engine);
    free(left); free(right); free(lines);
    return 0;
}

/* ------------------------------------------------------------------ mode: allocation audit */
static void strike_all(ow_pool* pool, ow_midi_event* ev, size_t n_eng, int restrike) {
    size_t k = 0;
    for (size_t e = 0; e < n_eng; ++e)
        for (int n = 33; n <= 96; ++n) {
            if (restrike) { ev[k].engine = (uint32_t)e; ev[k].type = 1; ev[k].note = (uint8_t)n; ev[k].reserved = 0; ev[k].value = 0.f; ++k; }
            ev[k].engine = (uint32_t)e; ev[k].type = 0; ev[k].note = (uint8_t)n; ev[k].reserved = 0;
            ev[k].value = (float)((40 + (37 * e) % 88) / 127.0); ++k;
        }
    ow_pool_midi(pool, ev, k);
}

static int cmd_audit(size_t n_eng, int n_renders) {
    const size_t L = 64;
    Dl_info di;
    ow_pool* pool = ow_pool_new(48000.0, n_eng, 0, OW_PREAMP_LEGACY8);
    ow_midi_event* ev;
    float* host;
    double peak = 0.0;
    if (!pool) { fprintf(stderr, "ow_pool_new failed: %s\n", ow_last_error()); return 3; }
    {   /* the library's load address, from the address of one of its functions as the library itself defines it */
        void* h = dlopen("libopenwurli_hip.so", RTLD_NOLOAD | RTLD_NOW);
        void* addr = h ? dlsym(h, "ow_pool_render") : NULL;
        if (!addr || !dladdr(addr, &di) || !strstr(di.dli_fname, "libopenwurli_hip")) { fprintf(stderr, "cannot locate libopenwurli_hip.so\n"); return 2; }
    }
    g_lib_base = di.dli_fbase;
    ow_pool_ensure_buffer_capacity(pool, L);
    ev = (ow_midi_event*)malloc(sizeof *ev * n_eng * 128);
    host = (float*)malloc(sizeof(float) * n_eng * L);
    if (!ev || !host) return 2;
    /* before the window: one strike + a few blocks (first launches load code objects inside the HIP runtime) */
    strike_all(pool, ev, n_eng, 0);
    for (int i = 0; i < 4; ++i) ow_pool_render(pool, host, L, L);
    g_window = 1;
    for (int i = 0; i < n_renders; ++i) {
        if (i == n_renders / 4) strike_all(pool, ev, n_eng, 1);                  /* whole-keyboard re-strike of every engine: 192 ops each */
        if (i == n_renders / 2)
            for (size_t e = 0; e < n_eng; e += 3) {                             /* setter retargets, sustain pedal, single notes */
                ow_engine* en = ow_pool_engine(pool, e);
                ow_engine_set_volume(en, 0.3 + 0.001 * (double)(e % 100)); ow_engine_set_tremolo_depth(en, 0.8); ow_engine_set_speaker_character(en, 0.4);
                ow_engine_set_sustain(en, 1); ow_engine_note_off(en, 60); ow_engine_note_on(en, 60, 0.9f);
            }
        ow_pool_render(pool, (i & 1) ? host : NULL, L, L);                       /* with and without the host copy */
    }
    g_window = 0;
    for (size_t i = 0; i < n_eng * L; ++i) if (fabs((double)host[i]) > peak) peak = fabs((double)host[i]);
    printf("{\"engines\": %zu, \"renders\": %d, \"lib_operator_new\": %lu, \"hip_alloc_calls\": %lu, \"peak\": %.6g, \"last_error\": \"%s\"}\n",
           n_eng, n_renders, g_lib_new, g_hip_alloc, peak, ow_last_error());
    ow_pool_free(pool);
    free(ev); free(host);
    return (g_lib_new == 0 && g_hip_alloc == 0 && peak > 1e-3) ? 0 : 1;
}

/* ------------------------------------------------------------------ mode: fault injection */
static int cmd_fault(size_t n_eng) {
    const size_t L = 256, stride = 300;
    ow_pool* pool = ow_pool_new(48000.0, n_eng, 0, OW_PREAMP_LEGACY8);
    float* host;
    int bad_rows = 0, untouched_ok = 1, recovered;
    double peak_before = 0.0, peak_after = 0.0;
    if (!pool) { fprintf(stderr, "ow_pool_new failed: %s\n", ow_last_error()); return 3; }
    host = (float*)malloc(sizeof(float) * n_eng * stride);
    if (!host) return 2;
    for (size_t e = 0; e < n_eng; ++e) { ow_engine* en = ow_pool_engine(pool, e); ow_engine_note_on(en, (uint8_t)(40 + e % 40), 0.8f); }
    for (int i = 0; i < 3; ++i) ow_pool_render(pool, host, stride, L);
    for (size_t e = 0; e < n_eng; ++e) for (size_t i = 0; i < L; ++i) if (fabs((double)host[e * stride + i]) > peak_before) peak_before = fabs((double)host[e * stride + i]);
    for (size_t i = 0; i < n_eng * stride; ++i) host[i] = 123.0f;               /* poison: a failing render must overwrite [e][0..L) of every row */
    ow_test_inject_render_faults(pool, 1);
    ow_pool_render(pool, host, stride, L);
    for (size_t e = 0; e < n_eng; ++e) {
        int row_bad = 0;
        for (size_t i = 0; i < L; ++i) if (host[e * stride + i] != 0.0f) row_bad = 1;
        for (size_t i = L; i < stride; ++i) if (host[e * stride + i] != 123.0f) untouched_ok = 0;   /* beyond len: not the library's to write */
        bad_rows += row_bad;
    }
    recovered = strstr(ow_last_error(), "injected fault") != NULL;
    ow_pool_render(pool, host, stride, L);
    for (size_t e = 0; e < n_eng; ++e) for (size_t i = 0; i < L; ++i) if (fabs((double)host[e * stride + i]) > peak_after) peak_after = fabs((double)host[e * stride + i]);
    printf("{\"engines\": %zu, \"rows_not_silent\": %d, \"padding_untouched\": %d, \"error_reported\": %d, \"peak_before\": %.6g, \"peak_after\": %.6g}\n",
           n_eng, bad_rows, untouched_ok, recovered, peak_before, peak_after);
    ow_pool_free(pool);
    free(host);
    return (bad_rows == 0 && untouched_ok && recovered && peak_before > 1e-3 && peak_after > 1e-3) ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc >= 7 && strcmp(argv[1], "render") == 0)
        return cmd_render(argv[2], argv[3], atof(argv[4]), (size_t)atol(argv[5]), (size_t)atol(argv[6]));
    if (argc >= 4 && strcmp(argv[1], "audit") == 0) return cmd_audit((size_t)atol(argv[2]), atoi(argv[3]));
    if (argc >= 3 && strcmp(argv[1], "fault") == 0) return cmd_fault((size_t)atol(argv[2]));
    fprintf(stderr, "usage: plugin_process render <script> <out.f32> <sr> <max_buffer> <n_buffers> | audit <n_engines> <n_renders> | fault <n_engines>\n");
    return 2;
}
