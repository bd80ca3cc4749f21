#!/bin/bash
# Round 6: the band-local seam of a persistent denoise kernel, measured in isolation (tools/ubench/band_seam.hip) -> gpurun_out/band_seam.txt
set -u
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/ubench/band_seam.hip -o /tmp/band_seam || exit 1
timeout 300 /tmp/band_seam | tee gpurun_out/band_seam.txt
