# round 3: mixer sum with four samples per thread and eight rows in flight, against HEAD (_base/); kernel trace of the AFC line with the kinds side by side
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_16; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q -k "mixer or mix" > $O/parity.log 2>&1; tail -3 $O/parity.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
run() { AIRBAND_HIP_LIB=$2 timeout 300 python bench.py $N $3 2>/dev/null | tail -1 > $O/$1.json; }
for round in 1 2; do
  run base_mix_$round $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--mixers 64 --force-dist"
  run new_mix_$round $L/libairband_hip.so "--mixers 64 --force-dist"
done
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 8 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_afc_forked -- python bench.py $K --afc 2 > $O/kt_afc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_mix -- python bench.py $K --mixers 64 --force-dist > $O/kt_mix.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -16 $f | cut -c1-180; done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_16"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], d["stage_ms"])
    except Exception as e: print(f, "ERR", e)
PY
