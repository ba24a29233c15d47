#!/bin/bash
# Register / LDS / scratch use of every gfx950 kernel of the library (compiler view).  usage: tools/kernel_resources.sh [filter]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-macro-redefined ${OW_SCHED--mllvm -amdgpu-sched-strategy=max-memory-clause} ${OW_HIPCC_EXTRA} \
  -Rpass-analysis=kernel-resource-usage --cuda-device-only -c -o /dev/null openwurli_amd/csrc/openwurli_hip.hip 2>&1 |
python3 -c '
import re,sys
txt=sys.stdin.read()
flt=sys.argv[1] if len(sys.argv)>1 else ""
cur=None; rows={}
for ln in txt.splitlines():
    m=re.search(r"Function Name: (\S+)",ln)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (\d+)",ln)
    if m and cur: rows[cur][m.group(1)]=int(m.group(2))
import subprocess
print("%-70s %5s %5s %5s %8s %5s %7s %6s" % ("kernel","VGPR","AGPR","SGPR","scratch","occ","LDS","vspill"))
for k,v in rows.items():
    name=subprocess.run(["c++filt",k],capture_output=True,text=True).stdout.strip().split("(")[0]
    if flt and flt not in name: continue
    print("%-70s %5s %5s %5s %8s %5s %7s %6s" % (name[:70], v.get("VGPRs"), v.get("AGPRs"), v.get("TotalSGPRs"), v.get("ScratchSize [bytes/lane]"), v.get("Occupancy [waves/SIMD]"), v.get("LDS Size [bytes/block]"), v.get("VGPRs Spill")))
' "$1"
