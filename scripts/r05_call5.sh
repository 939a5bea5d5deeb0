# Round 5, fifth GPU call: CS16 with the byte planes pulled apart once per landed step (-DAB_S16_PLANES): parity, then interleaved A/B; the wavefront FFT's tests on the
# product build (which now waits at every exchange).
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_call5; rm -rf $O; mkdir -p $O
EXP=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_s16planes.so
AIRBAND_HIP_LIB=$EXP timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_dropin_shim.py -q -m gpu -x -p no:cacheprovider -k "S16 or s16 or cs16 or classes" > $O/parity_planes.log 2>&1; tail -3 $O/parity_planes.log
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 20 --sample-format s16 --ring 1"
for rep in 1 2; do
  timeout 300 python bench.py $K 2>/dev/null | tail -n 1 > $O/cs16_product_$rep.json
  AIRBAND_HIP_LIB=$EXP timeout 300 python bench.py $K 2>/dev/null | tail -n 1 > $O/cs16_planes_$rep.json
done
timeout 600 python -m pytest tests/test_gpu_wavefront_fft.py tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "wavefront or fft_wave64 or SFMT_F32 or force_fft" > $O/fft_tests.log 2>&1; tail -3 $O/fft_tests.log
python - <<'PY'
import json, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05_call5"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j["ms_per_step"], j.get("stage_ms"), j.get("roofline",{}).get("frac"), j.get("verified_dongles"), j.get("build_info"))
    except Exception as e: print(f, "unreadable", e)
PY
