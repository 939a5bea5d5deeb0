# Round 6, call 10: the staging ring (review item 3 ii) -- parity first, then A/B against round 5's three buffers (-DAB_RING=0), policy variants, hops of 640 bytes with
# and without the ring; PMC fetch + TCP/TCC request counters; the ring under the pipelined mode.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c10; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -n 4 > $O/suite.log 2>&1; tail -3 $O/suite.log
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "configs1_1024_am or 200_mixed or 1000_mixed or 4096_mixed_splits2 or 1000_am_regrouped" > $O/scale.log 2>&1; tail -3 $O/scale.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 8"
for r in 1 2 3; do
  for l in ring noring headnt ringaux0; do
    lib=$L/libairband_hip.so; [ $l != ring ] && lib=$L/libairband_hip_exp_$l.so
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 40 2>$O/err_${l}_cfg3_$r.log | tail -1 > $O/${l}_cfg3_$r.json
  done
  for l in ring noring r640off; do
    lib=$L/libairband_hip.so; [ $l != ring ] && lib=$L/libairband_hip_exp_$l.so
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 40 --workload cfg2 --dongles 65536 2>/dev/null | tail -1 > $O/${l}_am65536_$r.json
  done
done
for l in ring noring; do
  lib=$L/libairband_hip.so; [ $l != ring ] && lib=$L/libairband_hip_exp_$l.so
  AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 40 --workload cfg4 2>/dev/null | tail -1 > $O/${l}_cfg4.json
  AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 200 --workload cfg2 2>/dev/null | tail -1 > $O/${l}_cfg2.json
  AIRBAND_HIP_LIB=$lib AIRBAND_HIP_DFT_EXTRA_LDS=9000 timeout 300 python bench.py $N --steps 40 --pipelined 2>/dev/null | tail -1 > $O/${l}_cfg3_pipelined_extra9000.json
  AIRBAND_HIP_LIB=$lib AIRBAND_HIP_DFT_EXTRA_LDS=12288 timeout 300 python bench.py $N --steps 40 --pipelined 2>/dev/null | tail -1 > $O/${l}_cfg3_pipelined_extra12288.json
  AIRBAND_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$l -- python bench.py $N --verify 0 --steps 3 --warmup 1 --dongles 32768 > $O/pmc_fetch_$l.log 2>&1
  AIRBAND_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_tcc_$l -- python bench.py $N --verify 0 --steps 3 --warmup 1 --dongles 32768 > $O/pmc_tcc_$l.log 2>&1
  AIRBAND_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$l -- python bench.py $N --verify 0 --steps 10 --warmup 2 > $O/kt_$l.log 2>&1
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c10"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["roofline"]["frac"], d["roofline"]["frac_read_only"], d["config"]["build_defines"])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
