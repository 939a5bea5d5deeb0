# LDS counters of the channelizer (hop 320: configs[2]; hop 640: 65 536 AM dongles)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; rm -rf $O; mkdir -p $O
P="--no-cpu-baseline --no-traffic --verify 0 --steps 2 --warmup 1"
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS"
timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/a -- python bench.py $P > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/b -- python bench.py $P --workload cfg2 --dongles 65536 > $O/b.log 2>&1
python - <<'PY'
import csv,glob,collections
for d in ("a","b"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r02o/%s/*/*counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            if "channelizer" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print("PMC",d,k,{c.replace("SQ_",""):"%.4g"%(sum(x)/len(x)) for c,x in v.items()})
PY
tail -2 $O/a.log
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
