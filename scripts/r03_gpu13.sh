# round 3: AFC groups read the shared table while at home; fast staging for hops that are not multiples of 16 bytes; stable groups for the plain NFM kind -- against HEAD (_base/)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_13; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
run() { AIRBAND_HIP_LIB=$2 timeout 300 python bench.py $N $3 2>/dev/null | tail -1 > $O/$1.json; }
for round in 1 2; do
  run base_cfg3_$round $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so
  run new_cfg3_$round $L/libairband_hip.so
  run base_afc_$round $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--afc 2"
  run new_afc_$round $L/libairband_hip.so "--afc 2"
  run base_2400k_$round $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--sample-rate 2400000"
  run new_2400k_$round $L/libairband_hip.so "--sample-rate 2400000"
done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_13"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], "chan %.3f demod %.3f"%(d["stage_ms"]["channelizer"], d["stage_ms"]["demod"]), "verified", d.get("verified_dongles"))
    except Exception as e: print(f, "ERR", e)
PY
