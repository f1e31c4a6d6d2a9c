#!/bin/bash
# Collect instruction-mix / stall counters of the photon kernel (separate PMC passes, kernel-trace only).
# usage: tools/pmc_profile.sh <outdir-under-gpurun_out> [bench args]
#        CASE=landsat tools/pmc_profile.sh <outdir>        profiles tools/run_case.py <CASE> instead of bench.py
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PROG=$R/bench.py
ARGS="--steps 1 --warmup 0 --photons 20000000 --no-cpu-baseline $@"
if [ -n "$CASE" ]; then PROG=$R/tools/run_case.py; ARGS="$CASE $@"; fi
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_IFETCH SQ_INSTS_FLAT" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $PROG $ARGS > $OUT/p$i.json 2> $OUT/p$i.err
  echo "pass $i done"
done
python3 - <<PY
import csv,glob
vals={}
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "photon_kernel" in r["Kernel_Name"] or "photon_pool_kernel" in r["Kernel_Name"]:
            vals[r["Counter_Name"]]=vals.get(r["Counter_Name"],0)+float(r["Counter_Value"])
            vals["_VGPR"]=r["VGPR_Count"]; vals["_SGPR"]=r["SGPR_Count"]; vals["_LDS"]=r["LDS_Block_Size"]; vals["_grid"]=r["Grid_Size"]
with open("$OUT/summary.txt","w") as o:
    for k in sorted(vals): o.write(f"{k} {vals[k]}\n")
print(open("$OUT/summary.txt").read())
PY
