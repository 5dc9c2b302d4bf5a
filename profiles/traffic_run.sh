set -u
for c in 3:k_inflate_batch 4:k_lzma_batch 5:k_deflate_batch; do
  cfg=${c%%:*}; k=${c##*:}
  for what in FETCH_SIZE WRITE_SIZE; do
    MZ_COLLECT_CONFIG=$cfg MZ_COLLECT_KERNEL=$k MZ_COLLECT_TIMEOUT=400 bash profiles/collect.sh traffic_cfg$cfg $what > gpurun_out/traffic_cfg${cfg}_$what.log 2>&1
  done
done
ls gpurun_out/traffic_cfg*/ | head -30
