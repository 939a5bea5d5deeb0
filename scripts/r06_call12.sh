# Round 6, call 12: CF32 kernel with its padding every 256 stream bytes (fragment and parking offsets as immediates: no spills at fft 2048) against the previous commit's library (_base).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c12; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "other_formats or afc or stage1 or chunk or cf32 or CF32 or lds" > $O/suite.log 2>&1; tail -3 $O/suite.log
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "F32" > $O/scale.log 2>&1; tail -3 $O/scale.log
F="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --sample-format f32 --ring 1 --dongles 32768"
for r in 1 2; do
  for l in new base; do
    lib=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip.so; [ $l = base ] && lib=$GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $F --steps 20 2>/dev/null | tail -1 > $O/${l}_f32_fft512_$r.json
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $F --steps 20 --fft-log 8 2>/dev/null | tail -1 > $O/${l}_f32_fft256_$r.json
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $F --steps 12 --fft-log 10 2>/dev/null | tail -1 > $O/${l}_f32_fft1024_$r.json
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $F --steps 8 --fft-log 11 2>/dev/null | tail -1 > $O/${l}_f32_fft2048_$r.json
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $F --steps 6 --fft-log 12 2>/dev/null | tail -1 > $O/${l}_f32_fft4096_$r.json
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $F --steps 20 --workload cfg2 --dongles 16384 2>/dev/null | tail -1 > $O/${l}_f32_am16384_$r.json
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c12"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["config"]["channelizer"], d["roofline"]["frac"])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
