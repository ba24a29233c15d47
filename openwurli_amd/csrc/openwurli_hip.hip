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

// The host side is ONE translation unit cut along its seams (round 6: 4 200 lines in one file before); every piece is included here, in
// dependency order, and nowhere else.
#include "host/host_base.inc"           // errors, device buffers, persistent host workers, latched switches
#include "host/host_types.inc"          // ow_engine / ow_pool: the host side of one engine and of a pool
#include "host/host_settle.inc"         // voice-pool mirror synchronisation, process-wide settled states (melange preamp / power amp, Twin-T)
#include "host/host_trajectory.inc"     // the shared Twin-T / CdS trajectory store (TremTraj), its helper thread, traj_acquire
#include "host/host_jobs_chain.inc"     // kernel choice for the job paths and the job chain (batch render, render-midi)
#include "host/host_pool.inc"           // tremolo phase groups, chain (re)initialisation, voice lists, render_range, post-render bookkeeping, pool life cycle
#include "host/api_pool.inc"            // C-ABI: library / pool entry points, host blocks, taps, trajectory control and persistence, MIDI bursts
#include "host/api_engines.inc"         // C-ABI: the WurliEngine API (engine.rs)
#include "host/api_test_hooks.inc"      // C-ABI of include/openwurli_hip_test.h: host-logic hooks, diagnostics, switches
#include "host/api_offline.inc"         // C-ABI: render_note, batch render, WAV writers, feature stage
#include "host/api_alias_audit.inc"     // C-ABI: the click-band alias audit (alias_audit.rs)
#include "host/api_midi_render.inc"     // C-ABI: `preamp-bench render-midi` (SMF reader, voice manager, chain)
