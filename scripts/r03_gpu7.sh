# round 3: A/B on ONE box -- the committed library (_base/) against the new channelizer loop (ladder waits, straight-line whole-tile stores,
# tiles software-pipelined inside the wave with three steps in flight) and the stage-2 variants (tail copy in rounds, AM at five waves, CTCSS chain in ranges)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_7; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
# parity first: the new channelizer loop and the chain ranges must be bit-/tolerance-clean before any number counts
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/parity_new.log 2>&1; tail -3 $O/parity_new.log
AIRBAND_HIP_LIB=$L/libairband_hip_exp_ch4am5.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stage2_bit_exact or full_slot_blocks or end_to_end" > $O/parity_ch4am5.log 2>&1; tail -3 $O/parity_ch4am5.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
run() { AIRBAND_HIP_LIB=$2 timeout 300 python bench.py $N $3 2>/dev/null | tail -1 > $O/$1.json; }
for round in 1 2; do
  run base_$round $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so
  run new_$round $L/libairband_hip.so
  run tc25_$round $L/libairband_hip_exp_tc25.so
  run am5_$round $L/libairband_hip_exp_am5.so
  run ch2_$round $L/libairband_hip_exp_ch2.so
  run ch4_$round $L/libairband_hip_exp_ch4.so
  run ch4am5_$round $L/libairband_hip_exp_ch4am5.so
done
run base_am $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--workload cfg2 --dongles 65536"
run new_am $L/libairband_hip.so "--workload cfg2 --dongles 65536"
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_7"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], "chan %.3f demod %.3f"%(d["stage_ms"]["channelizer"], d["stage_ms"]["demod"]), "verified", d.get("verified_dongles"))
    except Exception as e: print(f, "ERR", e)
PY
