#!/bin/bash
# Register / scratch / occupancy figures of the pass kernels as the compiler reports them (no GPU needed).
# usage: scripts/kernel_resources.sh [extra -D flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -fno-slp-vectorize --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" -x hip -c sunray_amd/csrc/kernels.hip -o /dev/null 2>&1 |
  python3 -c '
import sys, re
cur = None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur = t.split(":", 1)[1].strip()
    elif any(t.startswith(k) for k in ("TotalSGPRs:", "VGPRs:", "ScratchSize", "Occupancy", "SGPRs Spill", "VGPRs Spill")):
        cur += "  | " + t
if cur: print(cur)
' | grep -E "kernel" | sed "s/_ZN3srd[0-9]*//; s/EEvNS_8PassArgsE//; s/ILi/</"
