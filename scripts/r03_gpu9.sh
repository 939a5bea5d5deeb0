# round 3: the quiet-group path of stage 2 (four samples of a quiet wavefront as one basic block: AM kind, CTCSS front) against HEAD (_base/), one box
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_9; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/parity_new.log 2>&1; tail -3 $O/parity_new.log
AIRBAND_HIP_LIB=$L/libairband_hip_exp_fw3.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stage2_bit_exact or full_slot_blocks or end_to_end" > $O/parity_fw3.log 2>&1; tail -3 $O/parity_fw3.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
run() { AIRBAND_HIP_LIB=$2 timeout 300 python bench.py $N $3 2>/dev/null | tail -1 > $O/$1.json; }
for round in 1 2; do
  run base_$round $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so
  run q4_$round $L/libairband_hip.so
  run q4fw3_$round $L/libairband_hip_exp_fw3.so
done
run base_am $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--workload cfg2 --dongles 65536"
run q4_am $L/libairband_hip.so "--workload cfg2 --dongles 65536"
run base_cfg2 $GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so "--workload cfg2"
run q4_cfg2 $L/libairband_hip.so "--workload cfg2"
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 8 --warmup 2"
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -- python bench.py $K > $O/kt_serial.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -9 $f | cut -c1-200; done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_9"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], "chan %.3f demod %.3f"%(d["stage_ms"]["channelizer"], d["stage_ms"]["demod"]), "verified", d.get("verified_dongles"))
    except Exception as e: print(f, "ERR", e)
PY
