#!/bin/bash
# Round 5, call 10: instruction fetch of K1 (branches, fetches, I-cache hits / misses) on the probe
set -u
root=$PWD; out=$root/gpurun_out/c10; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $out/if$i -o pmc --output-format csv -- python $root/tests/perf_probe.py > $out/if$i.log 2>&1
  find $out/if$i -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|k_inflate_batch" "$1" > "$2"' _ {} $out/if$i.csv \;
  rm -rf $out/if$i
done
python3 - <<'PY'
import csv,collections,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/c10/if*.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(k, "%.4g"%(sum(v)/len(v)))
PY
