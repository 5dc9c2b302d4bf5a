#!/bin/bash
# Round 4, call 20: the whole GPU suite on HEAD with its full output (call 19's run died with a core dump and only its
# tail was kept)
set -u
mkdir -p gpurun_out/c20
python -c "import torch" 2>/dev/null
( timeout 1500 python -X faulthandler -m pytest tests -m gpu -v -x 2>&1 | grep -v amdgpu.ids ) > gpurun_out/c20/gputest_full.log 2>&1
grep -n "PASSED\|FAILED\|ERROR\|Fatal\|File \"" gpurun_out/c20/gputest_full.log | tail -40
tail -5 gpurun_out/c20/gputest_full.log
