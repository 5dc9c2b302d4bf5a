mkdir -p gpurun_out/c6
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_lzma_enc.py tests/test_gpu_xz.py -x -q 2>&1 | grep -v "^E   \s*$" | cut -c1-700 | tail -30 ) > gpurun_out/c6/tests.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_streams.py -x -q -s 2>&1 | grep -v "^E   \s*$" | cut -c1-700 | tail -30 ) > gpurun_out/c6/streams.log 2>&1
# HBM counters of the probe (20 000 x 64 KiB), the round-3 kernel against this round's: one pass per counter
export TMPDIR=/tmp
root=$PWD
cd /tmp
for tag in base final; do
  for c in FETCH_SIZE WRITE_SIZE; do
    MZHIP_LIB=$root/minizip-ng_amd/_build_ab_$tag/libmzhip.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $root/gpurun_out/c6/p_${tag}_$c -o pmc --output-format csv -- python $root/tests/perf_probe.py > $root/gpurun_out/c6/p_${tag}_$c.log 2>&1
    find $root/gpurun_out/c6/p_${tag}_$c -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|k_inflate_batch" "$1" > "$2"' _ {} $root/gpurun_out/c6/pmc_${tag}_$c.csv \;
    rm -rf $root/gpurun_out/c6/p_${tag}_$c
  done
done
cd $root
python3 - <<'PY'
import csv,glob
for f in sorted(glob.glob('gpurun_out/c6/pmc_*.csv')):
    v=[float(r['Counter_Value']) for r in csv.DictReader(open(f))]
    print(f.split('/')[-1], len(v), [round(x*1024/1e9,3) for x in v])
PY
tail -8 gpurun_out/c6/tests.log; tail -8 gpurun_out/c6/streams.log
