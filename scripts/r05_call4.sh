# Round 5, fourth GPU call: (1) what s_waitcnt lgkmcnt(0) at every wavefront-level exchange costs the exchange kernel (FORCE_FFT at configs[2], CF32 at fft 4096);
# (2) configs[1] at the old and the new signal start; (3) hops of an odd number of samples with ONE unaligned ds_read_b128 per fragment (-DAB_UNALIGNED_B128):
# parity, then 2.0 MS/s A/B.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_call4; rm -rf $O; mkdir -p $O
WAITS=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_sync_waits.so
UB=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_ub128.so
F="--no-cpu-baseline --no-traffic --no-verify-all --verify 2 --steps 5 --warmup 2"
for rep in 1 2; do
  AIRBAND_BENCH_FLAGS=4 timeout 300 python bench.py $F 2>/dev/null | tail -n 1 > $O/fft_product_$rep.json
  AIRBAND_BENCH_FLAGS=4 AIRBAND_HIP_LIB=$WAITS timeout 300 python bench.py $F 2>/dev/null | tail -n 1 > $O/fft_waits_$rep.json
done
timeout 300 python bench.py $F --sample-format f32 --ring 1 --dongles 8192 --fft-log 12 2>/dev/null | tail -n 1 > $O/f32_4096_product.json
AIRBAND_HIP_LIB=$WAITS timeout 300 python bench.py $F --sample-format f32 --ring 1 --dongles 8192 --fft-log 12 2>/dev/null | tail -n 1 > $O/f32_4096_waits.json
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4"
timeout 300 python bench.py $N --workload cfg2 --steps 400 2>/dev/null | tail -n 1 > $O/cfg2_start4.json
timeout 300 python bench.py $N --workload cfg2 --steps 400 --signal-start-batch 0 2>/dev/null | tail -n 1 > $O/cfg2_start0.json
AIRBAND_HIP_LIB=$UB timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "do_not_start_on_16 or 300_byte or (other_formats and (2_000_000 or 2000000 or 1_200_000 or 1200000 or 2400000 or 2_400_000))" > $O/parity_ub128.log 2>&1; tail -3 $O/parity_ub128.log
AIRBAND_HIP_LIB=$UB timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "other_formats" > $O/parity_ub128_formats.log 2>&1; tail -3 $O/parity_ub128_formats.log
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 30"
for rep in 1 2; do
  timeout 300 python bench.py $K --sample-rate 2000000 2>/dev/null | tail -n 1 > $O/r2000k_product_$rep.json
  AIRBAND_HIP_LIB=$UB timeout 300 python bench.py $K --sample-rate 2000000 2>/dev/null | tail -n 1 > $O/r2000k_ub128_$rep.json
done
python - <<'PY'
import json, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05_call4"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j["ms_per_step"], j.get("stage_ms"), j.get("roofline",{}).get("frac"), j.get("verified_dongles"), j.get("build_info"))
    except Exception as e: print(f, "unreadable", e)
PY
