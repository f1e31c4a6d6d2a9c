cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r03l; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export FUSION=1 I3RC_FUSED_GROUP_PHOTONS=50000000 REPS=1
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $GRAFT_REPO_ROOT/tools/fused_timing.py step16 1e6 300 > $O/kt.out 2>&1
python3 - <<'PY'
import csv, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03l"
f=glob.glob(O+"/kt/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
t0=min(int(r["Start_Timestamp"]) for r in rows)
for r in rows[-40:]:
    n=r["Kernel_Name"]
    short = "PHOTON" if "photon_kernel" in n else ("reduce_rep" if "reduce_replicas" in n else ("reduce_cnt" if "reduce_counters" in n else n[:30]))
    print("%-12s q%-3s start %10.3f ms  end %10.3f ms  dur %8.3f ms" % (short, r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
tail -1 $O/kt.out | cut -c1-200
