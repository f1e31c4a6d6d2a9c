#!/bin/bash
# Everything the judged evidence under profiles/ comes from, for one tag (e.g. r02): per workload
#   * rocprofv3 --kernel-trace --stats of `bench.py --config <w>` (kernel durations -> profiles/<tag>_<w>_kernel_stats.csv)
#   * the bench line itself                                         (-> profiles/<tag>_<w>_bench.json)
#   * the --pmc passes of tools/pmc_profile.sh                      (-> profiles/<tag>_<w>_pmc_summary.txt)
# and the issue-rate microbenchmark.  Runs on the GPU box; results land in gpurun_out/<tag>/ and are copied into profiles/
# by `tools/profile_round.sh collect <tag>` in the build container afterwards.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = collect ]; then
  TAG=$2; O=$R/gpurun_out/$TAG
  for w in step16 radar64_nadir landsat36 landsat119_7dir; do
    [ -f $O/bench_$w.json ] && cp $O/bench_$w.json $R/profiles/${TAG}_${w}_bench.json
    f=$(ls -t $O/stats_$w/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/profiles/${TAG}_${w}_kernel_stats.csv
    [ -f $O/pmc_$w/summary.txt ] && cp $O/pmc_$w/summary.txt $R/profiles/${TAG}_${w}_pmc_summary.txt
  done
  [ -f $O/issue_rate.txt ] && cp $O/issue_rate.txt $R/profiles/${TAG}_issue_rate_microbench.txt
  for f in atomic_rate call_overhead fused_timing driver_timing strong_scaling_proxy; do [ -f $O/$f.txt ] && grep -v "amdgpu.ids" $O/$f.txt > $R/profiles/${TAG}_$f.txt; done
  python3 $R/tools/pmc_to_json.py --collect $R/profiles/${TAG}_pmc.json $R/profiles/${TAG}_*_pmc_summary.txt > /dev/null
  ls $R/profiles | grep "^${TAG}_"
  exit 0
fi
TAG=$1; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
declare -A N=( [step16]=50000000 [radar64_nadir]=50000000 [landsat36]=100000000 [landsat119_7dir]=20000000 )
# counters first, so that the bench lines below quote THIS build's instruction mix and traffic (bench.py reads the newest profiles/*_pmc.json)
for w in ${WORKLOADS:-step16 radar64_nadir landsat36 landsat119_7dir}; do
  $R/tools/pmc_profile.sh $TAG/pmc_$w $w ${N[$w]} > $O/pmc_$w.log 2>&1 || echo "pmc $w failed"
  echo "$w counters done"
done
python3 $R/tools/pmc_to_json.py --collect $R/profiles/${TAG}_pmc.json $O/pmc_*/summary.txt > /dev/null && cp $R/profiles/${TAG}_pmc.json $O/pmc.json
for w in ${WORKLOADS:-step16 radar64_nadir landsat36 landsat119_7dir}; do
  # the profiled program is the rank itself (RANK set: bench.py's main() runs worker() at once), never bench.py's launcher --
  # a launcher hop after `--` would be traced only through the inherited preload, after the profiler library has touched the GPU
  ( export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 $R/bench.py --config $w --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_prof_$w.json 2> $O/bench_prof_$w.err ) || echo "stats $w failed"
  python3 $R/bench.py --config $w > $O/bench_$w.json 2> $O/bench_$w.err || echo "bench $w failed"
  echo "$w done: $(python3 -c "import json;j=json.load(open('$O/bench_$w.json'));print('%.3e photons/s'%j['value'], j['roofline']['kernel'])")"
done
[ -n "$SKIP_EXTRAS" ] && { echo "profile round $TAG done (workloads only)"; exit 0; }
[ -x $R/tools/microbench/issue_rate ] && $R/tools/microbench/issue_rate 5 > $O/issue_rate.txt 2>&1
[ -x $R/tools/microbench/atomic_rate ] && $R/tools/microbench/atomic_rate > $O/atomic_rate.txt 2>&1
# the batch loop: one call per batch, batches in flight, fused launches (tools/call_overhead.py, tools/fused_timing.py), the drivers end to end
cd $R
python3 tools/call_overhead.py step16 > $O/call_overhead.txt 2>&1
( for spec in "step16 1e5 1000" "step16 1e6 1000" "step16 1e7 30" "radar640 1e6 100" "landsat36 1e6 100" "landsat119 1e6 50"; do
    python3 tools/fused_timing.py $spec; I3RC_FUSED=0 python3 tools/fused_timing.py $spec; done ) > $O/fused_timing.txt 2>&1
bash tools/driver_timing.sh > $O/driver_timing.txt 2>&1
# strong scaling on one GPU: the shard of an 8-GPU run (1.25e7 photons per step) against the whole batch (1e8), one / two / three steps in flight
( for ov in 0 1 2; do python3 bench.py --photons 12500000 --steps 16 --warmup 2 --overlap $ov --no-cpu-baseline | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('shard of 1.25e7 photons x 16 steps, %d step(s) in flight: %.3e photons/s, %.3f ms per step, kernel %.3f ms' % (j['config']['steps_in_flight'], j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg']))"; done
  python3 bench.py --steps 5 --no-cpu-baseline | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('whole batch of 1e8 photons x 5 steps, 1 step in flight: %.3e photons/s, %.3f ms per step' % (j['value'], j['ms_per_step']))" ) > $O/strong_scaling_proxy.txt 2>&1
echo "profile round $TAG done"
