#!/bin/bash
# Build the gfx950 HIP library (C-ABI: include/openwurli_hip.h) and the CPU oracle.
set -e
cd "$(dirname "$0")"
mkdir -p openwurli_amd/lib
# -amdgpu-sched-strategy=max-memory-clause: the machine scheduler's strategy for every kernel (instruction order only, same arithmetic).
# Measured against the default strategy on one box, three runs each: k_preamp 6.69-6.76 ms against 6.86-6.96, everything else within
# noise (max-ilp: k_preamp 10.4 ms; iterative-minreg: 8.8 ms).  OW_SCHED= (empty) restores the compiler's default.
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=${OW_FP_CONTRACT:-off} -fPIC -shared -Wno-unused-value -Wno-macro-redefined \
    ${OW_SCHED--mllvm -amdgpu-sched-strategy=max-memory-clause} ${OW_HIPCC_EXTRA} -o openwurli_amd/lib/libopenwurli_hip.so openwurli_amd/csrc/openwurli_hip.hip
make -s -C oracle
