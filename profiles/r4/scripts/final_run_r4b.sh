#!/bin/bash
# Round 4, second evidence run (HEAD after the K3 / K6 work of the round's second session), one gpurun call from the repo root:
#   gpurun --timeout 2400 -- 'bash profiles/final_run_r4b.sh'      -> gpurun_out/final2/ ; profiles/harvest_r4b.sh copies
# what is judged into profiles/r4/.
set -u
mkdir -p gpurun_out/final2
python -c "import torch" 2>/dev/null
( timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -12 ) > gpurun_out/final2/gputest.log 2>&1
( python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > gpurun_out/final2/smoke.log 2>&1
# the default bench line (config 2 + other_configs + legs + cpu_baseline), then --stats and the two --pmc passes of config 2
bash profiles/collect.sh final2 all > gpurun_out/final2/collect.log 2>&1
# HBM counters of config 4 on HEAD (K3 changed this session; K1 and K4 did not: configs 3 and 5 keep the passes of 84a1b1f)
bash profiles/traffic_run.sh "4" > gpurun_out/final2/traffic_run.log 2>&1
cat gpurun_out/final2/gputest.log gpurun_out/final2/smoke.log; tail -c 1200 gpurun_out/final2/bench.log; ls gpurun_out/final2 gpurun_out/traffic_cfg4 2>/dev/null | head -40
