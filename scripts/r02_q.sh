# stage-2 schedule A/B over libraries: default run (kinds side by side), bench line + kernel trace
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; rm -rf $O; mkdir -p $O
for v in $VARIANTS; do
  [ "$v" = "base" ] && v=""
  export AIRBAND_HIP_LIB=$PWD/rtlsdr-airband_amd/libairband_hip$v.so
  timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-traffic --verify 8 2>/dev/null | tail -1 > $O/bench$v.json
  python -c "import json; j=json.load(open('$O/bench$v.json')); print('RESULT $v', j['ms_per_step'], {k:round(x,3) for k,x in j['stage_ms'].items()}, j.get('verified_dongles'), j['config']['build_defines'])"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$v -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 6 --warmup 2 > /dev/null 2>&1
  python - "$O/kt$v" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "airband" in r["Name"] and "siggen" not in r["Name"]: print("KT %-50s %8.3f"%(r["Name"].split("(")[0][-45:], float(r["AverageNs"])/1e6))
PY
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
