#!/bin/bash
# Compiler resource report (VGPRs / SGPRs / spills / occupancy) of every kernel: tools/resource_usage.sh [grep pattern]
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Rpass-analysis=kernel-resource-usage -o /tmp/libfk_ru.so /root/repo/farkle_ii_amd/csrc/farkle_hip.hip 2>&1 \
 | grep -E "Function Name|VGPRs:|SGPRs:|Spill|Occupancy|LDS Size|ScratchSize" | sed -e 's/^.*remark: [^ ]* *//' -e 's/ \[-Rpass.*$//' | paste -d' ' - - - - - - - - - | sed -e 's/Function Name: _ZN12_GLOBAL__N_1[0-9]*//' | grep -E "${1:-.}"
