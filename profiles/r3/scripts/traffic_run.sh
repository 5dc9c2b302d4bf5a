#!/bin/bash
# HBM counters of the bench configs other than the headline (bench.py --config N reports them as roofline.traffic):
#   gpurun --timeout 1500 -- 'bash profiles/traffic_run.sh "3 4"'   -> gpurun_out/traffic_cfgN/pmc_{fetch,write}_size.csv
#   python profiles/traffic_harvest.py r3 3 4                       -> profiles/r3/hbm_traffic_cfgN.json (+ the CSVs)
set -u
for cfg in ${1:-3 4 5}; do
  case $cfg in 3) k=k_inflate_batch;; 4) k=k_lzma;; 5) k=k_deflate_batch;; esac   # (config 4 = k_lzma_slot_batch + k_lzma_batch over what it gives back)
  for what in FETCH_SIZE WRITE_SIZE; do
    MZ_COLLECT_CONFIG=$cfg MZ_COLLECT_KERNEL=$k MZ_COLLECT_TIMEOUT=400 bash profiles/collect.sh traffic_cfg$cfg $what > gpurun_out/traffic_cfg${cfg}_$what.log 2>&1
  done
done
ls gpurun_out/traffic_cfg*/ | head -30
