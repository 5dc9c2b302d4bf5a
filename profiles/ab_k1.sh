#!/bin/bash
# A/B of K1 build variants.  Two steps, because hipcc time should not be paid on the GPU box:
#   profiles/ab_k1.sh build            (in the container) -> minizip-ng_amd/_build_ab_<tag>/libmzhip.so per variant
#   gpurun -- 'bash profiles/ab_k1.sh run'                  -> parity (tests/test_gpu_inflate.py) + tests/perf_probe.py per variant
# Variants: "<tag> <make variables>".  The .so files are git-ignored but travel with the gpurun snapshot.
set -u
VARIANTS=(
  "d8w8_k1 SLOTS=1"
  "d8w8_k2 SLOTS=2"
  "d8p6_w12_k1 POOL=6144 WPG=2 WAVES=3 SLOTS=1"
  "d6p4608_w14_k1 SPAN=6 POOL=4608 WPG=2 WAVES=4 SLOTS=1"
  "d6p3520_w16_k1 SPAN=6 POOL=3520 WPG=2 WAVES=4 SLOTS=1"
  "d6p3520_w16_k2 SPAN=6 POOL=3520 WPG=2 WAVES=4 SLOTS=2"
  "abl_d6_nonear SPAN=6 POOL=3520 WPG=2 WAVES=4 SLOTS=1 ABLATE=1"
  "abl_d6_walkemit SPAN=6 POOL=3520 WPG=2 WAVES=4 SLOTS=1 ABLATE=15"
)
root=$(cd "$(dirname "$0")/.." && pwd)
mode=${1:-run}
for v in "${VARIANTS[@]}"; do
  set -- $v
  tag=$1; shift
  dir=$root/minizip-ng_amd/_build_ab_$tag
  if [ "$mode" = build ]; then
    ( make -s -C "$root/minizip-ng_amd/csrc" OUT=../_build_ab_$tag "$@" > /tmp/ab_build_$tag.log 2>&1 && echo "built $tag ($*)" || echo "FAILED $tag" ) &
    while [ $(jobs -r | wc -l) -ge ${AB_JOBS:-8} ]; do sleep 1; done
  else
    [ -f "$dir/libmzhip.so" ] || { echo "$tag: not built"; continue; }
    echo "== $tag"
    ( cd "$root"; case $tag in abl_*) ;; *) MZHIP_LIB=$dir/libmzhip.so timeout 90 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -1;; esac
      MZHIP_LIB=$dir/libmzhip.so timeout 40 python tests/perf_probe.py 2>&1 | tail -1
      MZHIP_LIB=$dir/libmzhip.so timeout 40 python tests/perf_probe.py 512 200000 8192 2>&1 | tail -1 )
  fi
done
wait
