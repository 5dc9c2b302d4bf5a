set -u
root=$PWD; out=$root/gpurun_out/c7; mkdir -p $out
B=$root/minizip-ng_amd
{
echo "== K3 slot kernel: one full round of resident streams at 4 / 2 / 1 workgroups of four waves per CU (16 / 8 / 4 streams per CU)"
for v in "head 4096" "k3w2 2048" "k3w1 1024" "k3w1 4096" "k3w2 4096"; do set -- $v; lib=$B/_build/libmzhip.so; [ $1 != head ] && lib=$B/_build_ab_$1/libmzhip.so
  echo "-- $1, $2 entries"; MZHIP_LIB=$lib timeout 200 python tests/perf_codecs.py lzma $2 2>&1 | grep "LZMA decode"; done
echo "== K4 fast class: 20000 x 64 KiB at 4 / 2 / 1 workgroups per CU"
for v in head k4w2 k4w1; do lib=$B/_build/libmzhip.so; [ $v != head ] && lib=$B/_build_ab_$v/libmzhip.so
  echo "-- $v"; MZHIP_LIB=$lib timeout 200 python tests/perf_codecs.py deflate_only 2>&1 | grep "DEFLATE encode"; done
} > $out/occupancy.log 2>&1
cat $out/occupancy.log
for v in head k4w2 k4w1; do bash profiles/gpu.sh probe c7 $v k_deflate_batch python $root/tests/perf_codecs.py deflate_only; done
cat $out/req_summary.txt
