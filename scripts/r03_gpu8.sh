# round 3: A/B on ONE box -- pipelined channelizer loop (HEAD, _base/) against the same loop with the u8 -> int8 flip done once per staged byte in LDS
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_8; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/parity_new.log 2>&1; tail -3 $O/parity_new.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
run() { AIRBAND_HIP_LIB=$2 timeout 300 python bench.py $N $3 2>/dev/null | tail -1 > $O/$1.json; }
for round in 1 2 3; do
  run base_$round $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so
  run flip_$round $L/libairband_hip.so
done
run base_2400k $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--sample-rate 2400000"
run flip_2400k $L/libairband_hip.so "--sample-rate 2400000"
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_8"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], "chan %.3f demod %.3f"%(d["stage_ms"]["channelizer"], d["stage_ms"]["demod"]), "verified", d.get("verified_dongles"))
    except Exception as e: print(f, "ERR", e)
PY
