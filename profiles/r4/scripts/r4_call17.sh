#!/bin/bash
# Round 4, call 17: what K3's slot kernel issues per output byte, before and after the uniform branches / paired reads
# (instruction mix and busy cycles; a --pmc pass per counter group, nothing else traced)
set -u
export TMPDIR=/tmp
root=$PWD
mkdir -p gpurun_out/c17
python -c "import torch" 2>/dev/null
cd /tmp
for tag in k3_ubr0 default; do
  lib=$root/minizip-ng_amd/_build_ab_$tag/libmzhip.so
  [ $tag = default ] && lib=$root/minizip-ng_amd/_build/libmzhip.so
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"; do
    i=$((i+1))
    PYTHONPATH=$root MZHIP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $root/gpurun_out/c17/pmc_${tag}_$i -o pmc --output-format csv -- python $root/tests/perf_codecs.py lzma 4096 > $root/gpurun_out/c17/pmc_${tag}_$i.log 2>&1
    find $root/gpurun_out/c17/pmc_${tag}_$i -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|k_lzma_slot" "$1" > "$2"' _ {} $root/gpurun_out/c17/pmc_${tag}_$i.csv \;
    rm -rf $root/gpurun_out/c17/pmc_${tag}_$i
  done
done
cd $root
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/c17/pmc_*.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f, {k: "%.4g" % (sum(v) / len(v)) for k, v in acc.items()}, "launches", max(len(v) for v in acc.values()) if acc else 0)
PY
