#!/bin/bash
# The batch-loop evidence of a round (profiles/<tag>_fused_timing.txt, _lookahead_timing.txt, _driver_timing.txt,
# _strong_scaling_proxy.txt, _config_bench.txt): what tools/profile_round.sh also collects, without the counter passes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=$1; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( for spec in "step16 1e5 1000" "step16 1e6 1000" "step16 1e7 30" "step32 1e6 1000" "radar640 1e6 100" "landsat36 1e6 100" "landsat119 1e6 50"; do
    python3 tools/fused_timing.py $spec; I3RC_FUSED=0 python3 tools/fused_timing.py $spec; done ) > $O/fused_timing.txt 2>&1
( python3 tools/lookahead_timing.py step16 1e6 1000; python3 tools/lookahead_timing.py step32 1e6 1000; python3 tools/lookahead_timing.py landsat36 1e6 200 ) > $O/lookahead_timing.txt 2>&1
( bash tools/driver_timing.sh; bash tools/driver_timing.sh ) > $O/driver_timing.txt 2>&1
( for ov in 0 1 2; do python3 bench.py --photons 12500000 --steps 16 --warmup 2 --overlap $ov --no-cpu-baseline | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('shard of 1.25e7 photons x 16 steps, %d step(s) in flight: %.3e photons/s, %.3f ms per step, kernel %.3f ms' % (j['config']['steps_in_flight'], j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg']))"; done
  python3 bench.py --steps 5 --no-cpu-baseline | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('whole batch of 1e8 photons x 5 steps, 1 step in flight: %.3e photons/s, %.3f ms per step' % (j['value'], j['ms_per_step']))" ) > $O/strong_scaling_proxy.txt 2>&1
python3 tools/config_bench.py > $O/config_bench.txt 2>&1
for f in fused_timing lookahead_timing driver_timing strong_scaling_proxy config_bench; do grep -v "amdgpu.ids" $O/$f.txt > $O/$f.clean && mv $O/$f.clean $O/$f.txt; done
echo "batch loop evidence $TAG done"
