# Round 5, third GPU call: the tone kernel's recurrences fed by LDS broadcasts instead of a v_readlane per sample -- parity, then interleaved A/B against
# round 4's loops (-DAB_TONE_READLANE) and against the same code held to eight waves per SIMD (-DAB_TONE_WAVES8); then the exchange-kernel stress in the
# fuzz campaign's shape (small handles, 12 processes).
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_call3; rm -rf $O; mkdir -p $O
AIRBAND_FUZZ_SEEDS_GPU=120 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q -m gpu -x -p no:cacheprovider -k "stage2 or random_plans_on_the_gpu or end_to_end or golden or ctcss or full_slot" > $O/parity.log 2>&1; tail -3 $O/parity.log
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 30"
OLD=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_tone_readlane.so
W8=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_tone8.so
for rep in 1 2; do
  timeout 300 python bench.py $K 2>/dev/null | tail -n 1 > $O/bench_new_$rep.json
  AIRBAND_HIP_LIB=$OLD timeout 300 python bench.py $K 2>/dev/null | tail -n 1 > $O/bench_old_$rep.json
  AIRBAND_HIP_LIB=$W8 timeout 300 python bench.py $K 2>/dev/null | tail -n 1 > $O/bench_w8_$rep.json
done
python - <<'PY'
import json, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05_call3"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j["ms_per_step"], j.get("stage_ms"), j.get("build_info"))
    except Exception as e: print(f, "unreadable", e)
PY
P="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 6 --warmup 2"
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial_new -- python bench.py $P > $O/kt_new.log 2>&1
AIRBAND_HIP_LIB=$OLD AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial_old -- python bench.py $P > $O/kt_old.log 2>&1
for d in new old; do f=$(find $O/kt_serial_$d -name "*kernel_stats.csv" | head -1); echo "== $d"; grep -E "tone|demod|back|channelizer" $f | cut -d, -f1-4 | cut -c1-160; done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
bash scripts/r05_exchange_stress.sh 45 12 small
