#!/bin/bash
# A/B of K1 build variants.  Two steps, because hipcc time should not be paid on the GPU box:
#   profiles/ab_k1.sh build            (in the container) -> minizip-ng_amd/_build_ab_<tag>/libmzhip.so per variant
#   gpurun -- 'bash profiles/ab_k1.sh run'                  -> parity (tests/test_gpu_inflate.py) + tests/perf_probe.py per variant
# Variants: "<tag> <make variables>".  The .so files are git-ignored but travel with the gpurun snapshot.
set -u
VARIANTS=(
  "default  "
  "staged   STAGED=1"
  "staged_c8  STAGED=1 COMPACT=1"
  "staged_s6w6 SPAN=6 WAVES=6 STAGED=1 COMPACT=1"
  "staged_s6w5 SPAN=6 WAVES=5 STAGED=1 COMPACT=1"
  "staged2   STAGED=2"
  "staged2_pf STAGED=2 PREFETCH=1"
  "staged2_pf_s6w6 SPAN=6 WAVES=6 STAGED=2 COMPACT=1 PREFETCH=1"
  "staged_m2 STAGED=1 MLANES=2"
  "steploop SPAN=0 WAVES=8"
)
root=$(cd "$(dirname "$0")/.." && pwd)
mode=${1:-run}
for v in "${VARIANTS[@]}"; do
  set -- $v
  tag=$1; shift
  dir=$root/minizip-ng_amd/_build_ab_$tag
  if [ "$mode" = build ]; then
    make -s -C "$root/minizip-ng_amd/csrc" OUT=../_build_ab_$tag "$@" || exit 1
    echo "built $tag ($*)"
  else
    [ -f "$dir/libmzhip.so" ] || { echo "$tag: not built"; continue; }
    echo "== $tag"
    ( cd "$root" && MZHIP_LIB=$dir/libmzhip.so timeout 60 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -1
      MZHIP_LIB=$dir/libmzhip.so timeout 40 python tests/perf_probe.py 2>&1 | tail -1
      MZHIP_LIB=$dir/libmzhip.so timeout 40 python tests/perf_probe.py 512 200000 8192 2>&1 | tail -1 )
  fi
done
