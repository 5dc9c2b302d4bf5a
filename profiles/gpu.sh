#!/bin/bash
# profiles/gpu.sh -- the ONE script behind every measurement call of round 6 (rounds 4 and 5 kept a script per call:
# profiles/r4_call*.sh, r5_call*.sh, now under profiles/r4/scripts/ and profiles/r5/scripts/).
#
#   profiles/gpu.sh build  "<tag> [make variables]" ...      in the container: minizip-ng_amd/_build_ab_<tag>/libmzhip.so per variant
#                                                             (hipcc time is not paid on the GPU box; the .so files travel with gpurun)
#   gpurun -- 'bash profiles/gpu.sh ab <name> <tags...>'     parity (tests/test_gpu_inflate.py) + the 64 KiB and 8 KiB probes per variant
#   gpurun -- 'bash profiles/gpu.sh req <name> <tags...>'    + the L2's memory-side requests of k_inflate_batch on the 40 000 x 64 KiB
#                                                             probe per variant: read / write GB and L2 hit rate next to the ms
#   gpurun -- 'bash profiles/gpu.sh probe <name> <tag> <kernel substring> <cmd ...>'   two --pmc passes of any command on any build
#   gpurun -- 'bash profiles/gpu.sh evidence <name>'         bench.py (all configs) + rocprofv3 --kernel-trace --stats of the same command
#                                                             + the request counters of configs 2 - 5 on HEAD (hbm_traffic*.json inputs)
# Everything lands in gpurun_out/<name>/ ; what is to be judged is copied to profiles/r6/ afterwards (profiles/r6/README.md).
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
B=$root/minizip-ng_amd
mode=${1:-}; shift || true
export TMPDIR=/tmp

lib_of() { [ "$1" = head ] && echo $B/_build/libmzhip.so || echo $B/_build_ab_$1/libmzhip.so; }

req() { # out dir, tag, kernel substring, cmd...
  local out=$1 tag=$2 kern=$3; shift 3
  local i=0
  for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp -d $out/req_${tag}_$i -o pmc --output-format csv -- "$@" > $out/req_${tag}_$i.log 2>&1 )
    find $out/req_${tag}_$i -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|'"$kern"'" "$1" > "$2"' _ {} $out/req_${tag}_$i.csv \;
    rm -rf $out/req_${tag}_$i
  done
}

summary() { # out dir, tag, kernel substring, algorithmic bytes (0 = unknown)
  python3 - "$1" "$2" "$3" "${4:-0}" <<'EOF'
import csv, collections, sys
d, tag, kern, alg = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
acc, dur = collections.defaultdict(list), {}
for i in (1, 2):
    try:
        for r in csv.DictReader(open("%s/req_%s_%d.csv" % (d, tag, i))):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[(i, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    except FileNotFoundError:
        pass
m = {k: sum(v) / len(v) for k, v in acc.items()}
if not m:
    print("%-10s no counter rows" % tag); sys.exit(0)
n32, n64, n128 = m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_EA0_RDREQ_64B_sum", 0), m.get("TCC_EA0_RDREQ_128B_sum", 0)
w, w64 = m.get("TCC_EA0_WRREQ_sum", 0), m.get("TCC_EA0_WRREQ_64B_sum", 0)
rd, wr = 32 * n32 + 64 * n64 + 128 * n128, 32 * (w - w64) + 64 * w64
hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
ms = sum(dur.values()) / max(1, len(dur))
print("%-10s read %7.2f GB  write %6.2f GB  %s  L2 hit %4.1f %%  kernel %.3f ms under the counters (%d launches)" % (
    tag, rd / 1e9, wr / 1e9, ("= %.2f x algorithmic" % ((rd + wr) / alg)) if alg else "", 100.0 * hit / max(1.0, hit + miss), ms, len(dur)))
EOF
}

case "$mode" in
build)
  for v in "$@"; do
    set -- $v; tag=$1; shift
    args=(); for a in "$@"; do args+=("${a//%/ }"); done   # (a % inside a variable's value stands for a space: EXTRA=-DA=1%-DB=2)
    ( make -s -C "$B/csrc" OUT=../_build_ab_$tag "${args[@]}" > /tmp/ab_build_$tag.log 2>&1 && echo "built $tag ($*)" || { echo "FAILED $tag"; tail -5 /tmp/ab_build_$tag.log; } ) &
    while [ $(jobs -r | wc -l) -ge ${AB_JOBS:-6} ]; do sleep 1; done
  done
  wait ;;
ab|req)
  name=$1; shift
  out=$root/gpurun_out/$name; mkdir -p $out
  cd $root
  for tag in "$@"; do
    lib=$(lib_of $tag); [ -f "$lib" ] || { echo "$tag: not built"; continue; }
    {
      echo "== $tag"
      case $tag in abl_*) ;; *) MZHIP_LIB=$lib timeout 120 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -1;; esac
      for rep in 1 2; do MZHIP_LIB=$lib timeout 60 python tests/perf_probe.py 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; done
      MZHIP_LIB=$lib timeout 60 python tests/perf_probe.py 512 200000 8192 2>&1 | grep -v '^rep [01]\|amdgpu.ids'
    } >> $out/probe.log 2>&1
  done
  cat $out/probe.log
  if [ "$mode" = req ]; then
    for tag in "$@"; do
      lib=$(lib_of $tag); [ -f "$lib" ] || continue
      MZHIP_LIB=$lib req $out $tag k_inflate_batch python $root/tests/perf_probe.py 2048 40000
    done
    # algorithmic bytes of the probe: 40 000 x (65 536 + c), c from the probe's own ratio line
    alg=$(python3 -c "print(int(40000 * 65536 * (1 + 0.2966)))")
    for tag in "$@"; do summary $out $tag k_inflate_batch $alg; done | tee $out/req_summary.txt
  fi ;;
probe)
  name=$1; tag=$2; kern=$3; shift 3
  out=$root/gpurun_out/$name; mkdir -p $out
  MZHIP_LIB=$(lib_of $tag) req $out $tag "$kern" "$@"
  summary $out $tag "$kern" 0 | tee -a $out/req_summary.txt ;;
evidence)
  name=$1; shift
  out=$root/gpurun_out/$name; mkdir -p $out
  cd $root
  timeout 1500 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
  tail -c 600 $out/bench.err
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats -o k --output-format csv -- python $root/bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-other-configs > $out/bench_under_rocprof.json 2> $out/stats.err )
  find $out/stats -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
  rm -rf $out/stats
  export HARVEST_OUT=$out HARVEST_COMMIT=$(cat $root/.commit 2>/dev/null || echo unknown)
  for cfg in 2 3 4 5; do
    case $cfg in 2|3) kern=k_inflate_batch; hk=k_inflate_batch;; 4) kern=k_lzma; hk=k_lzma+;; 5) kern=k_deflate_batch; hk=k_deflate_batch;; esac
    req $out cfg$cfg $kern python $root/bench.py --config $cfg --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-other-configs
    summary $out cfg$cfg $kern 0
    # hbm_traffic*.json of this very build: entries, entry size and algorithmic bytes from the JSON line the run printed
    python3 - $out cfg$cfg $cfg "$hk" <<'PY'
import json, subprocess, sys, os
out, tag, cfg, hk = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
line = None
for ln in open(os.path.join(out, "req_%s_1.log" % tag), errors="replace"):
    if ln.startswith("{") and '"roofline"' in ln:
        line = json.loads(ln)
if line:
    c = line["config"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(out)))
    subprocess.run([sys.executable, os.path.join(root, "profiles", "req_harvest.py"), "r6", out, tag, str(cfg), str(c.get("entries_total", 0)), str(c.get("entry_bytes", 0)),
                    str(line["roofline"]["algorithmic_bytes_per_launch"]), hk])
PY
  done | tee $out/req_summary.txt
  head -c 1500 $out/bench.json ;;
*)
  sed -n 2,16p "$0" ;;
esac
