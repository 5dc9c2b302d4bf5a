set -u
root=$PWD; out=$root/gpurun_out/prio4; mkdir -p $out
B=$root/minizip-ng_amd
{
for v in head d1 d2 d3 d4 d5 d6 d7 d8 d9 head; do lib=$B/_build/libmzhip.so; [ $v != head ] && lib=$B/_build_ab_$v/libmzhip.so
  echo "-- $v"; for r in 1 2; do MZHIP_LIB=$lib timeout 200 python tests/perf_codecs.py deflate_only 2>&1 | grep "DEFLATE encode"; done; done
for v in head z1 z2 z3 head; do lib=$B/_build/libmzhip.so; [ $v != head ] && lib=$B/_build_ab_$v/libmzhip.so
  echo "-- $v"; MZHIP_LIB=$lib timeout 300 python tests/perf_codecs.py lzma 4096 2>&1 | grep "LZMA decode"; done
} > $out/k34.log 2>&1
cat $out/k34.log
