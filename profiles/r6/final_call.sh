# the round's last call: evidence (bench + rocprof stats + request counters of configs 2 - 5) on the final build, its traffic
# files put where bench.py looks for them, then the GPU suite, smoke, the default bench once more, and one rank's share of a
# table cut eight ways (12 500 entries)
set -u
root=$PWD; out=$root/gpurun_out/fin5; mkdir -p $out
bash profiles/gpu.sh evidence ev_fin > $out/evidence.out 2>&1
cp gpurun_out/ev_fin/hbm_traffic*.json profiles/r6/
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $out/gputest_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke >> $out/gputest_final.log
python bench.py --steps 20 --warmup 5 > $out/bench_final.json 2> $out/bench_final.err
python bench.py --steps 20 --warmup 5 --entries 12500 --no-legs --no-cpu-baseline --no-other-configs > $out/bench_12500.json 2> $out/bench_12500.err
cat $out/gputest_final.log; tail -c 1200 $out/evidence.out; head -c 400 $out/bench_final.json; echo; head -c 300 $out/bench_12500.json
