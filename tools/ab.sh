#!/bin/bash
# A/B timing on the GPU box: tools/ab.sh "<workload> <photons>" lib1 lib2 ...   (lib = name of a variant built by
# tools/variant_bench.py build, "default" for the library itself, or a path); env EVTHR / LITHR are passed on.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CASE=$1; shift
for v in "$@"; do
  case $v in
    default) unset I3RC_LIB;;
    /*) export I3RC_LIB=$v;;
    *) export I3RC_LIB=$R/i3rc-monte-carlo-model_amd/csrc/libi3rc_hip_var_$v.so;;
  esac
  echo "$v: $(timeout -k 10 200 python3 $R/tools/run_case.py $CASE 2>&1 | tail -1)"
done
