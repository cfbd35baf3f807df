#!/bin/bash
# builds the 4-wave GEMM probe (extra -D flags as arguments); output tools/probes/_bin/g4w_probe[suffix]
cd "$(dirname "$0")" && mkdir -p _bin
OUT=_bin/g4w_probe${SUFFIX}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVC_4W_STAMP "$@" g4w_probe.hip -o $OUT && ls -la $OUT
