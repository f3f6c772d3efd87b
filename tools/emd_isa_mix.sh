#!/bin/bash
# Vector-instruction mix of emd_cost_kernel's inner loops (the constants of bench.py::EMD_LOOPS), from the ISA hipcc emits.
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -I$HERE/pdgn_amd/csrc \
    -I$HERE/include $HERE/pdgn_amd/csrc/structural.hip -o /tmp/structural.s 2>/dev/null
awk '/^_Z15emd_cost_kernel/,/uses_flat_scratch/' /tmp/structural.s > /tmp/emd.s
for L in $(grep -n "Inner Loop Header: Depth=[34]" /tmp/emd.s | cut -d: -f1); do
    awk -v s=$L 'NR>=s{print; if ($1 ~ /s_cbranch/ && NR>s+20) exit}' /tmp/emd.s |
        awk -v L=$L '{c[$1]++} END{printf "loop at line %d: ", L; for(k in c) if (k ~ /^v_|^ds_/) printf "%s=%d ", k, c[k]; print ""}'
done
