# Round 6, call 22: CF32 at hops of an odd number of 16-byte units (2.4 MS/s) without the per-lane offset arrays (layout 3) against the round's earlier library (_base: layout 0).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c22; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "other_formats or afc or stage1 or cf32 or CF32 or lds" > $O/suite.log 2>&1; tail -n 2 $O/suite.log
F="--no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 4 --sample-format f32 --ring 1 --dongles 32768 --sample-rate 2400000"
FB="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --sample-format f32 --ring 1 --dongles 32768 --sample-rate 2400000"
for r in 1 2; do
  for fl in 9 11 12; do
    timeout 300 python bench.py $F --steps 8 --fft-log $fl 2>$O/err_new_$fl.log | tail -1 > $O/new_f32_2400k_fft${fl}_$r.json
    AIRBAND_HIP_LIB=$GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so timeout 300 python bench.py $FB --steps 8 --fft-log $fl 2>$O/err_base_$fl.log | tail -1 > $O/base_f32_2400k_fft${fl}_$r.json
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c22"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["config"]["channelizer"], d["roofline"]["frac"])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
