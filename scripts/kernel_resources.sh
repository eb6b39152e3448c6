#!/bin/bash
# Register, spill, LDS and occupancy figures of every kernel in device/kernels.hip (compiler remarks).
# Usage: kernel_resources.sh [extra hipcc flags]
cd "$(dirname "$0")/../raytracer_amd/csrc"
/opt/rocm/bin/hipcc -O3 -ffp-contract=off -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -I../../include -Ifront -Idevice \
  -Rpass-analysis=kernel-resource-usage "$@" -c device/kernels.hip -o /tmp/kernels_res.o 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|SGPRs:" | sed -e 's/.*remark: //' |
  awk '/Function Name/ {printf "\n%s", $0; next} {printf " | %s", $0} END {print ""}'
