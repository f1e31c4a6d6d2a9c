#!/bin/bash
# Everything under profiles/<tag>_* comes from here, from ONE output directory (gpurun_out/<tag>/) -- round 3 had two scripts with two
# directories, and collecting one of them put older measurements over newer ones.
#   tools/evidence_round.sh <tag> counters   (GPU box) per BASELINE workload: the --pmc passes (tools/pmc_profile.sh), rocprofv3
#                                            --kernel-trace --stats of `bench.py --config <w>`, the bench line itself
#   tools/evidence_round.sh <tag> general    (GPU box) the same for the general-class workloads (several components, irregular grid, gridded surface)
#   tools/evidence_round.sh <tag> absorbing  (GPU box) the same for the absorbing I3RC cases and the tool-chain domain
#   tools/evidence_round.sh <tag> loop       (GPU box) the batch loop and the rest: fused / look-ahead / moments timing, the
#                                            drivers end to end, the strong-scaling proxy, config_bench, the phase profile,
#                                            call overhead, the issue-rate and atomic-rate microbenchmarks
#   tools/evidence_round.sh <tag> collect    (build container, after both) gpurun_out/<tag>/ -> profiles/<tag>_*
# The two GPU parts fit a gpurun call each (about 12 and 8 minutes); neither removes what the other wrote.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; PART=$2; O=$R/gpurun_out/$TAG
WL="step16 radar64_nadir landsat36 landsat119_7dir"
# round 5: the problems beyond the common class -- several components, an irregular x / y grid, a gridded surface (tools/workloads.py)
GL="landsat119_gas landsat119_gas_7dir landsat119_irregular_7dir landsat119_brdfgrid_7dir landsat36_aerosol_gas"
# ... the I3RC cases' absorbing versions (omega = 0.99) and the LES stratocumulus + Rayleigh domain the reference's tool chain wrote
AL="step16_absorbing landsat36_absorbing landsat119_absorbing les_stcu_rayleigh les_stcu_rayleigh_2dir"
clean() { grep -v "amdgpu.ids" "$1" > "$1.clean" && mv "$1.clean" "$1"; }
case $PART in
counters)
  mkdir -p $O; cd /tmp; export TMPDIR=/tmp
  declare -A N=( [step16]=50000000 [radar64_nadir]=50000000 [landsat36]=100000000 [landsat119_7dir]=20000000 )
  # counters first, so that the bench lines below quote THIS build's instruction mix and traffic (bench.py reads the newest profiles/*_pmc.json)
  for w in ${WORKLOADS:-$WL}; do
    $R/tools/pmc_profile.sh $TAG/pmc_$w $w ${N[$w]} > $O/pmc_$w.log 2>&1 || echo "pmc $w failed"
    echo "$w counters done"
  done
  python3 $R/tools/pmc_to_json.py --collect $R/profiles/${TAG}_pmc.json $O/pmc_*/summary.txt > /dev/null && cp $R/profiles/${TAG}_pmc.json $O/pmc.json
  for w in ${WORKLOADS:-$WL}; do
    # the profiled program is the rank itself (RANK set: bench.py's main() runs worker() at once), never bench.py's launcher
    ( export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 $R/bench.py --config $w --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_prof_$w.json 2> $O/bench_prof_$w.err ) || echo "stats $w failed"
    python3 $R/bench.py --config $w > $O/bench_$w.json 2> $O/bench_$w.err || echo "bench $w failed"
    echo "$w done: $(python3 -c "import json;j=json.load(open('$O/bench_$w.json'));print('%.3e photons/s'%j['value'], j['roofline']['kernel'])")"
  done
  echo "evidence $TAG: counters done";;
general|absorbing)
  # the same three things -- counter passes, rocprofv3 kernel stats of the bench, the bench line -- for the general-class workloads
  # (absorbing: for the I3RC cases' absorbing versions and the domain of the reference's tool chain)
  mkdir -p $O; cd /tmp; export TMPDIR=/tmp
  declare -A N=( [landsat119_gas]=50000000 [landsat119_gas_7dir]=10000000 [landsat119_irregular_7dir]=10000000 [landsat119_brdfgrid_7dir]=10000000 [landsat36_aerosol_gas]=50000000
                 [step16_absorbing]=50000000 [landsat36_absorbing]=50000000 [landsat119_absorbing]=50000000 [les_stcu_rayleigh]=50000000 [les_stcu_rayleigh_2dir]=20000000 )
  [ $PART = absorbing ] && GL=$AL
  for w in ${WORKLOADS:-$GL}; do
    PASSES="1 2 9 10" $R/tools/pmc_profile.sh $TAG/pmc_$w $w ${N[$w]} > $O/pmc_$w.log 2>&1 || echo "pmc $w failed"
    ( export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 $R/bench.py --config $w --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof_$w.json 2> $O/bench_prof_$w.err ) || echo "stats $w failed"
    python3 $R/bench.py --config $w --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || echo "bench $w failed"
    echo "$w done: $(python3 -c "import json;j=json.load(open('$O/bench_$w.json'));print('%.3e photons/s'%j['value'], j['roofline']['kernel'])")"
  done
  [ $PART = absorbing ] && { echo "evidence $TAG: absorbing done"; exit 0; }
  # ... and each of them on the GENERAL kernels (what ran them until round 4) and on the round-4 place of their field, one launch each
  ( export REPEAT=3
    for w in landsat119_gas landsat119_gas_7dir landsat119_irregular_7dir landsat119_brdfgrid_7dir; do
      n=${N[$w]}; python3 $R/tools/run_case.py $w $n | tail -1; KERNEL=general python3 $R/tools/run_case.py $w $n | tail -1
      case $w in *gas*) GRID=bricks KERNEL=general python3 $R/tools/run_case.py $w $n | tail -1;; esac
    done
    python3 $R/tools/run_case.py landsat119 50000000 | tail -1; python3 $R/tools/run_case.py landsat119_7dir 10000000 | tail -1 ) > $O/general_kernels.txt 2>&1; clean $O/general_kernels.txt   # (REPEAT=3: the last of three launches, not a process's first)
  # ... and their batch loops: fused launches of the widened class against one launch per batch
  ( cd $R; for spec in "landsat36_gas 1e6 60" "landsat119_gas 1e6 40" "landsat119_gas_7dir 1e6 20" "landsat119_irregular_7dir 1e6 20" "landsat119_brdfgrid_7dir 1e6 20" "landsat36_aerosol_gas 1e6 50" "les_stcu_rayleigh 1e6 50" "step16_absorbing 1e6 300" "landsat36_absorbing 1e6 100"; do
      python3 tools/fused_timing.py $spec; I3RC_FUSED=0 python3 tools/fused_timing.py $spec; done ) > $O/fused_wide.txt 2>&1; clean $O/fused_wide.txt
  echo "evidence $TAG: general done";;
loop)
  mkdir -p $O; cd $R
  ( for spec in "step16 1e5 1000" "step16 1e6 1000" "step32 1e6 1000" "radar640 1e6 100" "landsat36 1e6 100" "landsat119 1e6 50" "radar64_nadir 1e6 100" "landsat119_7dir 1e6 28"; do
      python3 tools/fused_timing.py $spec; I3RC_FUSED=0 python3 tools/fused_timing.py $spec; done ) > $O/fused_timing.txt 2>&1; clean $O/fused_timing.txt
  ( for spec in "landsat36 1e6 100" "landsat119 1e6 50" "step16 1e6 300" "radar64_nadir 1e6 100" "landsat119_7dir 1e6 28" "step16 1e3 20000"; do python3 tools/moments_timing.py $spec; done ) > $O/moments_timing.txt 2>&1; clean $O/moments_timing.txt
  ( python3 tools/lookahead_timing.py step16 1e6 1000; python3 tools/lookahead_timing.py step32 1e6 1000; python3 tools/lookahead_timing.py landsat36 1e6 200
    python3 tools/lookahead_timing.py radar64_nadir 1e6 200 ) > $O/lookahead_timing.txt 2>&1; clean $O/lookahead_timing.txt
  ( bash tools/driver_timing.sh; bash tools/driver_timing.sh ) > $O/driver_timing.txt 2>&1; clean $O/driver_timing.txt
  # strong scaling on one GPU: the shard of an 8-GPU run (1.25e7 photons per step) against the whole batch (1e8), one / two / three steps in flight
  ( for ov in 0 1 2; do python3 bench.py --photons 12500000 --steps 16 --warmup 2 --overlap $ov --no-cpu-baseline | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('shard of 1.25e7 photons x 16 steps, %d step(s) in flight: %.3e photons/s, %.3f ms per step, kernel %.3f ms' % (j['config']['steps_in_flight'], j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg']))"; done
    python3 bench.py --steps 5 --no-cpu-baseline | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('whole batch of 1e8 photons x 5 steps, 1 step in flight: %.3e photons/s, %.3f ms per step' % (j['value'], j['ms_per_step']))" ) > $O/strong_scaling_proxy.txt 2>&1; clean $O/strong_scaling_proxy.txt
  python3 tools/config_bench.py > $O/config_bench.txt 2>&1; clean $O/config_bench.txt
  python3 tools/phase_profile.py --build step16 5e7 radar64_nadir 5e7 landsat36 5e7 landsat119_7dir 2e7 > $O/phase_profile.txt 2>&1; clean $O/phase_profile.txt
  python3 tools/call_overhead.py step16 > $O/call_overhead.txt 2>&1; clean $O/call_overhead.txt
  [ -x tools/microbench/issue_rate ] && tools/microbench/issue_rate 5 > $O/issue_rate.txt 2>&1
  [ -x tools/microbench/atomic_rate ] && tools/microbench/atomic_rate > $O/atomic_rate.txt 2>&1
  echo "evidence $TAG: loop done";;
collect)
  [ -f $O/general_kernels.txt ] && cp $O/general_kernels.txt $R/profiles/${TAG}_general_kernels.txt
  [ -f $O/fused_wide.txt ] && cp $O/fused_wide.txt $R/profiles/${TAG}_fused_wide.txt
  for w in $WL $GL $AL; do
    [ -f $O/bench_$w.json ] && cp $O/bench_$w.json $R/profiles/${TAG}_${w}_bench.json
    f=$(ls -t $O/stats_$w/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/profiles/${TAG}_${w}_kernel_stats.csv
    [ -f $O/pmc_$w/summary.txt ] && cp $O/pmc_$w/summary.txt $R/profiles/${TAG}_${w}_pmc_summary.txt
  done
  [ -f $O/issue_rate.txt ] && cp $O/issue_rate.txt $R/profiles/${TAG}_issue_rate_microbench.txt
  for f in atomic_rate call_overhead fused_timing moments_timing lookahead_timing driver_timing strong_scaling_proxy config_bench phase_profile; do
    [ -f $O/$f.txt ] && cp $O/$f.txt $R/profiles/${TAG}_$f.txt; done
  python3 $R/tools/pmc_to_json.py --collect $R/profiles/${TAG}_pmc.json $R/profiles/${TAG}_*_pmc_summary.txt > /dev/null
  ls -la --time-style=+%H:%M $R/profiles | grep " ${TAG}_";;
*) echo "usage: tools/evidence_round.sh <tag> counters|general|absorbing|loop|collect"; exit 2;;
esac
