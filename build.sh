#!/bin/bash
# Build the gfx950 HIP library (C-ABI: include/openwurli_hip.h) and the CPU oracle.
set -e
cd "$(dirname "$0")"
mkdir -p openwurli_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=${OW_FP_CONTRACT:-off} -fPIC -shared -Wno-unused-value -Wno-macro-redefined \
    ${OW_HIPCC_EXTRA} -o openwurli_amd/lib/libopenwurli_hip.so openwurli_amd/csrc/openwurli_hip.hip
make -s -C oracle
