# Round 6, call 18: line groups permuted per batch in front of one-wavefront workgroups (AIRBAND_HIP_REGROUP=3) against slot order (0) and channels in lockstep workgroups (1).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c18; rm -rf $O; mkdir -p $O
AIRBAND_HIP_REGROUP=3 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_dropin_shim.py -m gpu -x -q -n 4 > $O/suite_rg3.log 2>&1; tail -n 2 $O/suite_rg2.log
AIRBAND_HIP_REGROUP=3 timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "regroup" > $O/scale_rg3.log 2>&1; tail -n 2 $O/scale_rg2.log
N="--no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 8 --steps 40"
for r in 1 2 3; do
  for m in 0 3 1; do
    AIRBAND_HIP_REGROUP=$m timeout 300 python bench.py $N 2>$O/err_rg${m}_$r.log | tail -1 > $O/rg${m}_cfg3_$r.json
  done
done
for m in 0 3 1; do
  AIRBAND_HIP_REGROUP=$m timeout 300 python bench.py $N --workload cfg4 2>/dev/null | tail -1 > $O/rg${m}_cfg4.json
  AIRBAND_HIP_REGROUP=$m timeout 300 python bench.py $N --key-on-s 0.15 2>/dev/null | tail -1 > $O/rg${m}_duty10.json
  AIRBAND_HIP_REGROUP=$m timeout 300 python bench.py $N --key-on-s 0.15 --workload cfg4 2>/dev/null | tail -1 > $O/rg${m}_duty10_cfg4.json
  AIRBAND_HIP_REGROUP=$m timeout 300 python bench.py $N --workload cfg2 --dongles 65536 2>/dev/null | tail -1 > $O/rg${m}_am65536.json
  AIRBAND_HIP_REGROUP=$m timeout 300 python bench.py $N --dongles 49152 2>/dev/null | tail -1 > $O/rg${m}_49152.json
  AIRBAND_HIP_REGROUP=$m timeout 300 python bench.py $N --dongles 16384 2>/dev/null | tail -1 > $O/rg${m}_16384.json
done
for m in 0 3; do
  AIRBAND_HIP_REGROUP=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_rg$m -- python bench.py $N --verify 0 --steps 8 --warmup 2 > $O/kt_rg$m.log 2>&1
  AIRBAND_HIP_REGROUP=$m AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial_rg$m -- python bench.py $N --verify 0 --steps 8 --warmup 2 > $O/kt_serial_rg$m.log 2>&1
  AIRBAND_HIP_REGROUP=$m timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_rg$m -- python bench.py $N --verify 0 --steps 3 --warmup 1 --dongles 32768 > $O/pmc_fetch_rg$m.log 2>&1
  AIRBAND_HIP_REGROUP=$m AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_rg$m -- python bench.py $N --verify 0 --steps 2 --warmup 1 > $O/pmc_sq_rg$m.log 2>&1
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c18"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), "open", d.get("open_fraction", {}).get("mean"), d["config"].get("stage2_regrouped"))
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
