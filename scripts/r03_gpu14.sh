# round 3: AFC spectrum beside stage 1 (parity + line), and the order experiment "AM kind behind the split chain"
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_14; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "afc or end_to_end" > $O/parity_afc.log 2>&1; tail -3 $O/parity_afc.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
run() { AIRBAND_HIP_LIB=$2 timeout 300 python bench.py $N $3 2>/dev/null | tail -1 > $O/$1.json; }
for round in 1 2 3; do
  run new_cfg3_$round $L/libairband_hip.so
  run amlast_cfg3_$round $L/libairband_hip_exp_amlast.so
  run new_afc_$round $L/libairband_hip.so "--afc 2"
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_14"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], "chan %.3f demod %.3f"%(d["stage_ms"]["channelizer"], d["stage_ms"]["demod"]), "verified", d.get("verified_dongles"))
    except Exception as e: print(f, "ERR", e)
PY
