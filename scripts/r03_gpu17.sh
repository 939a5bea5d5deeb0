# round 3: re-tune kernel as one wavefront per work item (AFC line), parity of the AFC tests
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_17; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "afc or mixer or other_formats" > $O/parity.log 2>&1; tail -3 $O/parity.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
run() { AIRBAND_HIP_LIB=$2 timeout 300 python bench.py $N $3 2>/dev/null | tail -1 > $O/$1.json; }
for round in 1 2; do
  run new_cfg3_$round $L/libairband_hip.so
  run base_afc_$round $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--afc 2"
  run new_afc_$round $L/libairband_hip.so "--afc 2"
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_17"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], "chan %.3f demod %.3f"%(d["stage_ms"]["channelizer"], d["stage_ms"]["demod"]), "verified", d.get("verified_dongles"))
    except Exception as e: print(f, "ERR", e)
PY
