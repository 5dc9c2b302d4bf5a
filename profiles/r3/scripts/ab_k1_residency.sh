#!/bin/bash
# K1 residency study: the shipped kernel at 16 / 12 / 8 waves per CU (LDS padding only changes
# how many workgroups fit a CU), timed and -- PROF=1 builds -- with the per-section cycle sums of a wave.  A section whose
# per-wave cycles do not move with the residency is latency-bound (more waves = more throughput); one whose cycles scale
# with the residency is bound by a shared resource (issue slots, LDS, memory).
#   make -C minizip-ng_amd/csrc OUT=../_build_ab_<tag> [LDSPAD=3328|10240] [PROF=1]     (in the container)
#   gpurun -- 'bash profiles/ab_k1_residency.sh'
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
for tag in ${AB_TAGS:-base pad12 pad8 prof pad12_prof pad8_prof}; do
  lib=$root/minizip-ng_amd/_build_ab_$tag/libmzhip.so
  [ -f "$lib" ] || { echo "$tag: not built"; continue; }
  echo "== $tag"
  MZHIP_LIB=$lib timeout 60 python tests/perf_probe.py 2>&1 | grep -v '^rep [01]\|amdgpu.ids'
done
