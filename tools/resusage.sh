#!/bin/bash
# compact per-kernel resource usage (VGPR/AGPR/spill/LDS/occupancy) of the gfx950 build
cd "$(dirname "$0")/../helm_amd/csrc"
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage -c helm_hip.hip -o /tmp/helm_hip_res.o 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//; s/ \[-Rpass.*//' |
  awk '/Function Name/{if(l)print l; l=$3; next}{gsub(/^ +/,""); l=l" | "$0}END{print l}' | grep "${1:-k_}"
