cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r03e; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export FUSION=1 I3RC_FUSED_GROUP_PHOTONS=1000000000 REPS=1
for spec in "radar640 1e8 1" "radar640 5e7 2" "landsat36 1e8 1" "landsat36 1e7 10"; do
  set -- $spec; tag=$1_$3
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/a_$tag -- python3 $GRAFT_REPO_ROOT/tools/fused_timing.py $1 $2 $3 > $O/a_$tag.out 2>&1 || echo fail a $tag
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/b_$tag -- python3 $GRAFT_REPO_ROOT/tools/fused_timing.py $1 $2 $3 > $O/b_$tag.out 2>&1 || echo fail b $tag
  tail -1 $O/a_$tag.out | cut -c1-200
done
python3 - <<'PY'
import csv, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03e"
for d in sorted(glob.glob(O+"/[ab]_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc={}
        for r in csv.DictReader(open(f)):
            if "photon_kernel" in r["Kernel_Name"] and "Batch" in r["Kernel_Name"]:
                key=(r["Dispatch_Id"], r["Counter_Name"])
                acc[key]=acc.get(key,0)+float(r["Counter_Value"])
        disp=sorted({k[0] for k in acc}, key=int)
        for dd in disp[-1:]:
            print(os.path.basename(d), "dispatch", dd, {k[1]: "%.4g"%v for k,v in acc.items() if k[0]==dd})
PY
