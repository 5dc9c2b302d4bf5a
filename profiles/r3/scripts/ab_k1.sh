#!/bin/bash
# A/B of K1 build variants.  Two steps, because hipcc time should not be paid on the GPU box:
#   profiles/ab_k1.sh build            (in the container) -> minizip-ng_amd/_build_ab_<tag>/libmzhip.so per variant
#   gpurun -- 'bash profiles/ab_k1.sh run'                  -> parity (tests/test_gpu_inflate.py) + tests/perf_probe.py per variant
# Variants: "<tag> <make variables>".  The .so files are git-ignored but travel with the gpurun snapshot.
set -u
VARIANTS=(
  "base"
  "sel EXTRA=-DMZ_TOKEN_SELECT=1"
  "pre2 PRELIT=2"
  "chunked EXTRA=-DMZ_REC_CHUNKED=1"
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
      MZHIP_LIB=$dir/libmzhip.so timeout 40 python tests/perf_probe.py 2>&1 | grep -v '^rep [01]\|amdgpu.ids'
      MZHIP_LIB=$dir/libmzhip.so timeout 40 python tests/perf_probe.py 512 200000 8192 2>&1 | grep -v '^rep [01]\|amdgpu.ids' )
  fi
done
wait
