# round 3: handles without a CTCSS chain run their first fused kind on the caller's stream (no fork / join)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_15; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/parity.log 2>&1; tail -3 $O/parity.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 400"
run() { AIRBAND_HIP_LIB=$2 timeout 300 python bench.py $N $3 2>/dev/null | tail -1 > $O/$1.json; }
for round in 1 2; do
  run base_cfg2_$round $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--workload cfg2"
  run new_cfg2_$round $L/libairband_hip.so "--workload cfg2"
done
run base_am $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--workload cfg2 --dongles 65536 --steps 40"
run new_am $L/libairband_hip.so "--workload cfg2 --dongles 65536 --steps 40"
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_15"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], "chan %.3f demod %.3f"%(d["stage_ms"]["channelizer"], d["stage_ms"]["demod"]), "verified", d.get("verified_dongles"))
    except Exception as e: print(f, "ERR", e)
PY
