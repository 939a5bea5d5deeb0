# Round 6, call 21: stable-group block for the NFM + lowpass kind (product) against the kind as it was (-DAB_NO_LP_STABLE4).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c21; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -n 4 > $O/suite.log 2>&1; tail -n 2 $O/suite.log
N="--no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 8 --steps 40"
for r in 1 2 3; do
  for l in prod nolp4; do
    lib=$L/libairband_hip.so; [ $l != prod ] && lib=$L/libairband_hip_exp_$l.so
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N 2>$O/err_${l}_$r.log | tail -1 > $O/${l}_cfg3_$r.json
  done
done
for l in prod nolp4; do
  lib=$L/libairband_hip.so; [ $l != prod ] && lib=$L/libairband_hip_exp_$l.so
  AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --workload cfg4 2>/dev/null | tail -1 > $O/${l}_cfg4.json
  AIRBAND_HIP_LIB=$lib AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial_$l -- python bench.py $N --verify 0 --steps 8 --warmup 2 > $O/kt_serial_$l.log 2>&1
  AIRBAND_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$l -- python bench.py $N --verify 0 --steps 8 --warmup 2 > $O/kt_$l.log 2>&1
  AIRBAND_HIP_LIB=$lib AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq_$l -- python bench.py $N --verify 0 --steps 2 --warmup 1 > $O/pmc_sq_$l.log 2>&1
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c21"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["config"]["build_defines"])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
