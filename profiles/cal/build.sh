#!/bin/bash
# profiles/cal/libcal.so: the calibration kernels (git-ignored like every built .so; travels to the GPU box)
cd "$(dirname "$0")" && hipcc --offload-arch=gfx950 -O2 -shared -fPIC cal_kernels.hip -o libcal.so
