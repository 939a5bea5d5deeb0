# round 4: what limits the matrix-core channelizer at fft >= 1024 (window pieces on cooperating waves): matrix-pipe busy cycles, LDS, waits
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/f1024; rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|SQ_BUSY_CU|SQ_WAIT_INST|SQ_ACTIVE_INST" | cut -c1-140 > $O/counters_offered.txt; head -40 $O/counters_offered.txt
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 2 --warmup 1 --ring 1"
for L in 9 10 11 12; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_I8 --output-format csv -d $O/pmc_a_$L -- python bench.py $K --fft-log $L > $O/pmc_a_$L.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $O/pmc_b_$L -- python bench.py $K --fft-log $L > $O/pmc_b_$L.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/f1024"
for d in sorted(glob.glob(O+"/pmc_*_*")):
    if not os.path.isdir(d): continue
    agg=collections.defaultdict(list); dur=[]
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "channelizer_dft" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    print(os.path.basename(d), "ms %.2f"%(sum(dur)/max(1,len(dur))), {k:"%.4g"%(sum(v)/len(v)) for k,v in agg.items()})
PY
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
