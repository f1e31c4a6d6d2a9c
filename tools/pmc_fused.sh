cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r03z; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export REPS=1
for spec in "step16 1e6 300" "landsat36 1e6 100"; do
  set -- $spec; tag=$1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/a_$tag -- python3 $GRAFT_REPO_ROOT/tools/fused_timing.py $1 $2 $3 > $O/a_$tag.out 2>&1 || echo fail a $tag
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/b_$tag -- python3 $GRAFT_REPO_ROOT/tools/fused_timing.py $1 $2 $3 > $O/b_$tag.out 2>&1 || echo fail b $tag
done
python3 - <<'PY'
import csv, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03z"
for tag, photons in (("step16", 3e8), ("landsat36", 1e8)):
    tot={}
    for d in ("a_"+tag, "b_"+tag):
        for f in glob.glob(O+"/"+d+"/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "photon_kernel" in r["Kernel_Name"] and "Batch" in r["Kernel_Name"]:
                    tot[r["Counter_Name"]]=tot.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    if tot:
        print("%s fused (%g photons): VALU instr / photon %.1f, SALU %.1f, FLAT (atomics) %.2f, lane occupancy %.3f, wait-any %.1f %%, wait-inst %.1f %% of wave cycles" % (
            tag, photons, tot["SQ_INSTS_VALU"]/photons, tot["SQ_INSTS_SALU"]/photons, tot["SQ_INSTS_FLAT"]/photons, tot["SQ_THREAD_CYCLES_VALU"]/(64*tot["SQ_ACTIVE_INST_VALU"]),
            100*tot["SQ_WAIT_ANY"]/tot["SQ_WAVE_CYCLES"], 100*tot["SQ_WAIT_INST_ANY"]/tot["SQ_WAVE_CYCLES"]))
PY
