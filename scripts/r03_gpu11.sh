# round 3: stable groups (quiet groups extended through the timed squelch states) against HEAD (_base/ = quiet groups only), one box
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_11; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_dropin_shim.py -m gpu -x -q > $O/parity_new.log 2>&1; tail -3 $O/parity_new.log
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "1024 or 200 or 1000 or 4096" > $O/scale_small.log 2>&1; tail -3 $O/scale_small.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
run() { AIRBAND_HIP_LIB=$2 timeout 300 python bench.py $N $3 2>/dev/null | tail -1 > $O/$1.json; }
for round in 1 2; do
  run base_cfg3_$round $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so
  run st4_cfg3_$round $L/libairband_hip.so
  run base_cfg2_$round $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--workload cfg2 --steps 200"
  run st4_cfg2_$round $L/libairband_hip.so "--workload cfg2 --steps 200"
done
run base_am $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--workload cfg2 --dongles 65536"
run st4_am $L/libairband_hip.so "--workload cfg2 --dongles 65536"
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_11"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], "chan %.3f demod %.3f"%(d["stage_ms"]["channelizer"], d["stage_ms"]["demod"]), "verified", d.get("verified_dongles"))
    except Exception as e: print(f, "ERR", e)
PY
