# Round 6, call 6: merged tone banks (product) against the previous commit's library (_base); regrouping over handle sizes; the new fabric test.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c6; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fabric.py -m gpu -x -q -n 4 > $O/suite.log 2>&1; tail -3 $O/suite.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 8 --warmup 2"
for round in 1 2 3; do
  for l in new base; do
    lib=$L/libairband_hip.so; [ $l = base ] && lib=$GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N 2>$O/err_${l}_$round.log | tail -1 > $O/${l}_cfg3_$round.json
  done
done
for d in 8192 16384 24576 32768 49152; do
  for r in 0 1; do
    timeout 300 python bench.py $N --regroup $r --dongles $d 2>/dev/null | tail -1 > $O/size${d}_rg$r.json
  done
done
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -- python bench.py $K > $O/kt_serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py $K > $O/kt.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c6"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["config"].get("stage2_regrouped"))
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
