# Round 6, call 15: how much of the window-piece sizes' launch time is bytes in flight -- half the product's (experiment builds -DAB_NP_INFLIGHT=1 / 2) and fewer workgroups per CU.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c15; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4"
for r in 1 2; do
  for l in prod inflight1 inflight2; do
    lib=$L/libairband_hip.so; [ $l != prod ] && lib=$L/libairband_hip_exp_$l.so
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 30 --fft-log 10 2>$O/err_${l}_$r.log | tail -1 > $O/${l}_fft1024_$r.json
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 20 --fft-log 11 2>/dev/null | tail -1 > $O/${l}_fft2048_$r.json
  done
  for x in 2100 16000; do
    AIRBAND_HIP_DFT_EXTRA_LDS=$x timeout 300 python bench.py $N --steps 30 --fft-log 10 2>/dev/null | tail -1 > $O/prod_fft1024_extra${x}_$r.json
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c15"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["config"]["build_defines"])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/final_suite
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $GRAFT_REPO_ROOT/gpurun_out/final_suite/gpu_suite.log 2>&1; grep -E "passed|failed" $GRAFT_REPO_ROOT/gpurun_out/final_suite/gpu_suite.log | tail -n 2
