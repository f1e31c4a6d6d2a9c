#!/bin/bash
# rocprofv3 counter passes of one workload's photon kernel (separate --pmc passes, kernel-trace only; FETCH_SIZE and
# WRITE_SIZE each in a pass of their own, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).
# usage: tools/pmc_profile.sh <outdir-under-gpurun_out> <workload> [photons]      (workloads: tools/workloads.py)
#        PASSES="1 2 9 10" limits the passes (1 instruction counts, 2 activity / lane occupancy, 3 VALU mix, 4 f64 / LDS,
#        5 waits, 6 GRBM, 9 FETCH, 10 WRITE)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=$R/gpurun_out/$1; CASE=$2; shift; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
declare -A SETS
SETS[1]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT"
SETS[2]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
SETS[3]="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_LDS_ATOMIC"
SETS[4]="SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH"
SETS[5]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
SETS[6]="GRBM_GUI_ACTIVE"
SETS[9]="FETCH_SIZE"
SETS[10]="WRITE_SIZE"
for i in ${PASSES:-1 2 3 4 5 6 9 10}; do
  rocprofv3 --pmc ${SETS[$i]} --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/run_case.py $CASE "$@" > $OUT/p$i.out 2> $OUT/p$i.err || { echo "pass $i failed"; tail -5 $OUT/p$i.err; exit 1; }
  echo "pass $i done: $(tail -1 $OUT/p$i.out)"
done
python3 $R/tools/pmc_to_json.py --summarise $OUT $CASE
