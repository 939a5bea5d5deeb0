# A/B of experiment libraries (VARIANTS="base _exp_x ...") + instruction-fetch counters of the stage-2 kernels run alone
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; rm -rf $O; mkdir -p $O
for v in $VARIANTS; do
  [ "$v" = "base" ] && v=""
  export AIRBAND_HIP_LIB=$PWD/rtlsdr-airband_amd/libairband_hip$v.so
  timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-traffic --verify 4 2>/dev/null | tail -1 > $O/bench$v.json
  python -c "import json; j=json.load(open('$O/bench$v.json')); print('RESULT $v', j['ms_per_step'], {k:round(x,3) for k,x in j['stage_ms'].items()}, j.get('verified_dongles'), j['config']['build_defines'])"
  AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$v -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 6 --warmup 2 > /dev/null 2>&1
  grep -h "demod_kernel\|tone_kernel\|back_kernel" $O/kt$v/*/*kernel_stats.csv | cut -d, -f1,4 | sed 's/"void airband:://; s/(airband::DemodArgs[^"]*"//; s/"airband:://'
done
unset AIRBAND_HIP_LIB
if [ -n "$PMC" ]; then
P="--no-cpu-baseline --no-traffic --verify 0 --steps 2 --warmup 1"
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $O/pmc_a -- python bench.py $P > $O/pmc_a.log 2>&1
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc_b -- python bench.py $P > $O/pmc_b.log 2>&1
python - <<'PY'
import csv,glob,collections
for d in ("pmc_a","pmc_b"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r02i/%s/*/*counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            if "airband" in r["Kernel_Name"] and "siggen" not in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print("PMC",d,k,{c:"%.4g"%(sum(x)/len(x)) for c,x in v.items()})
PY
tail -3 $O/pmc_a.log $O/pmc_b.log
fi
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
