#!/bin/bash
# dev: register / scratch / LDS numbers of the kernels of one source file:  bash tools/dev/res.sh icp [pattern] [extra flags]
F=${1:-icp}; PAT=${2:-.}; EXTRA=${3:-}
cd "$(dirname "$0")/../../threecrate_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-result $EXTRA -Rpass-analysis=kernel-resource-usage -c $F.hip -o /tmp/res_$F.o 2>&1 | \
  grep -E "error|warning:|Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size|VGPRs Spill|TotalSGPRs" | \
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | awk '/Function Name/ {printf "\n%s\n  ", $0; next} /error|warning/ {print; next} {printf "%s | ", $0}' | grep -A1 -E "$PAT|error|warning" | grep -v "^--"
echo
