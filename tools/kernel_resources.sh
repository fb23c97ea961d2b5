#!/bin/bash
# VGPRs / scratch / occupancy of every kernel, from hipcc's own resource-usage remarks (no GPU needed):
#   bash tools/kernel_resources.sh > profiles/<tag>_kernel_resources.txt
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
printf '#include <hip/hip_runtime.h>\n#include <cstdint>\n#include "qe_types.h"\n#include "qe_kernels.hip"\n' > $tmp/t.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only -O3 -std=c++17 -c -Rpass-analysis=kernel-resource-usage \
  -I$root/include -I$root/quicked_amd/csrc $tmp/t.hip -o $tmp/t.o 2>&1 \
  | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: //; s/ \[-Rpass-analysis=kernel-resource-usage\]//' \
  | awk '/Function Name/ { if (line) print line; cmd = "c++filt " $3; cmd | getline name; close(cmd); line = name; next } { gsub(/^ +/, ""); line = line " | " $0 } END { print line }'
rm -rf $tmp
