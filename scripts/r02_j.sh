# parity subset + timing + SQ instruction counters of the stage-2 kernels run alone, for the library named by $LIBV ("" = product)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; rm -rf $O; mkdir -p $O
[ -n "$LIBV" ] && export AIRBAND_HIP_LIB=$PWD/rtlsdr-airband_amd/libairband_hip$LIBV.so
if [ -n "$TESTS" ]; then timeout 900 python -m pytest tests -x -q -m gpu -k "$TESTS" 2>&1 | tail -5; fi
timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-traffic --verify 8 2>/dev/null | tail -1 > $O/bench.json
python -c "import json; j=json.load(open('$O/bench.json')); print('RESULT', j['ms_per_step'], {k:round(x,3) for k,x in j['stage_ms'].items()}, j.get('verified_dongles'), j['config']['build_defines'])"
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 6 --warmup 2 > /dev/null 2>&1
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 2 --warmup 1 > $O/pmc.log 2>&1
python - <<'PY'
import csv,glob,collections
for f in glob.glob("gpurun_out/r02j/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "airband" in r["Name"] and "siggen" not in r["Name"]: print("KT %-50s %8.3f"%(r["Name"].split("(")[0][-45:], float(r["AverageNs"])/1e6))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r02j/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "airband" in r["Kernel_Name"] and "siggen" not in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print("PMC",k,{c.replace("SQ_",""):"%.4g"%(sum(x)/len(x)) for c,x in v.items()})
PY
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
