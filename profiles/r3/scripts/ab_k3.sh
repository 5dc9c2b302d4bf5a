#!/bin/bash
# K3 residency experiment: how much would more streams per CU buy?  The literal model (12 KiB of the
# 15.6 KiB per wave) is what keeps K3 at 10 waves per CU.  Upper bound of any scheme that moves part of it out of LDS:
# streams coded with lc = 0 need 0x300 literal probabilities only, so a build with MZ_LZMA_MAX_LCLP=0 (5.2 KiB per wave)
# decodes them at 12 / 16 / 20 waves per CU -- with NO out-of-LDS access at all.
#   profiles/ab_k3.sh build       (container)   |   gpurun -- 'bash profiles/ab_k3.sh run'
set -u
VARIANTS=(
  "k3_base"
  "k3_l0_r10 EXTRA=-DMZ_LZMA_MAX_LCLP=0"
  "k3_l0_r12 EXTRA=-DMZ_LZMA_MAX_LCLP=0%-DMZ_LZMA_RESIDENT=12u"
  "k3_l0_w4_r16 EXTRA=-DMZ_LZMA_MAX_LCLP=0%-DMZ_LZMA_WAVES=4%-DMZ_LZMA_RESIDENT=16u"
  "k3_l0_w5_r20 EXTRA=-DMZ_LZMA_MAX_LCLP=0%-DMZ_LZMA_WAVES=5%-DMZ_LZMA_RESIDENT=20u"
  "k3_l0_w6_r24 EXTRA=-DMZ_LZMA_MAX_LCLP=0%-DMZ_LZMA_WAVES=6%-DMZ_LZMA_RESIDENT=24u"
)
root=$(cd "$(dirname "$0")/.." && pwd)
mode=${1:-run}
for v in "${VARIANTS[@]}"; do
  set -- $v
  tag=$1; shift
  dir=$root/minizip-ng_amd/_build_ab_$tag
  if [ "$mode" = build ]; then
    args=(); for a in "$@"; do args+=("${a//%/ }"); done
    ( make -s -C "$root/minizip-ng_amd/csrc" OUT=../_build_ab_$tag "${args[@]}" > /tmp/ab_build_$tag.log 2>&1 && echo "built $tag" || echo "FAILED $tag" ) &
    while [ $(jobs -r | wc -l) -ge ${AB_JOBS:-8} ]; do sleep 1; done
  else
    [ -f "$dir/libmzhip.so" ] || { echo "$tag: not built"; continue; }
    echo "== $tag"
    ( cd "$root"; LZMA_LC=0 MZHIP_LIB=$dir/libmzhip.so timeout 120 python tests/perf_codecs.py lzma ${K3_ENTRIES:-6144} 2>&1 | grep "LZMA decode" )
  fi
done
wait
