set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/f32pmc; rm -rf $O; mkdir -p $O
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 2 --warmup 1 --ring 1 --sample-format f32 --dongles 32768"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/a -- python bench.py $K > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/b -- python bench.py $K > $O/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c -- python bench.py $K > $O/c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/d -- python bench.py $K > $O/d.log 2>&1
python - <<'PY'
import csv, glob, collections, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/f32pmc"
for d in sorted(glob.glob(O+"/?")):
    agg=collections.defaultdict(list); dur=[]
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "channelizer_f32" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    print(os.path.basename(d), "ms %.2f"%(sum(dur)/max(1,len(dur))), {k:"%.4g"%(sum(v)/len(v)) for k,v in agg.items()})
PY
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
