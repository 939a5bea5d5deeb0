# Round 6, call 5: workgroup-local regrouping with lockstep (AIRBAND_HIP_FLAG_REGROUP, second form) -- parity, then A/B against slot order; masked delay on top; nt default against aux 0.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c5; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
AIRBAND_HIP_REGROUP=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_dropin_shim.py tests/test_gpu_fabric.py -m gpu -x -q -n 4 > $O/suite_regrouped.log 2>&1; tail -3 $O/suite_regrouped.log
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "plans or regroup" > $O/new_tests.log 2>&1; tail -3 $O/new_tests.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 8 --warmup 2"
for round in 1 2 3; do
  for l in base md nont; do
    lib=$L/libairband_hip.so; [ $l != base ] && lib=$L/libairband_hip_exp_$l.so
    for r in 0 1; do
      [ $l = nont ] && [ $r = 1 ] && continue
      AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --regroup $r 2>$O/err_${l}_rg${r}_$round.log | tail -1 > $O/${l}_rg${r}_cfg3_$round.json
    done
  done
done
for r in 0 1; do
  timeout 300 python bench.py $N --regroup $r --key-on-s 0.15 2>/dev/null | tail -1 > $O/base_rg${r}_duty10.json
  timeout 300 python bench.py $N --regroup $r --workload cfg2 --dongles 65536 2>/dev/null | tail -1 > $O/base_rg${r}_am65536.json
  timeout 300 python bench.py $N --regroup $r --workload cfg4 2>/dev/null | tail -1 > $O/base_rg${r}_cfg4.json
  timeout 300 python bench.py $N --regroup $r --workload cfg2 --steps 400 2>/dev/null | tail -1 > $O/base_rg${r}_cfg2.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_rg$r -- python bench.py $K --regroup $r > $O/kt_rg$r.log 2>&1
  AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial_rg$r -- python bench.py $K --regroup $r > $O/kt_serial_rg$r.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_rg$r -- python bench.py --no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 3 --warmup 1 --dongles 32768 --regroup $r > $O/pmc_rg$r.log 2>&1
  AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH --output-format csv -d $O/pmc_sq_serial_rg$r -- python bench.py --no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 2 --warmup 1 --regroup $r > $O/pmc_sq_rg$r.log 2>&1
done
timeout 600 python bench.py --no-cpu-baseline --no-traffic --verify 8 --steps 20 --regroup 1 2>$O/err_rg1_verify_all.log | tail -1 > $O/rg1_cfg3_verify_all.json
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c5"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), "open", d.get("open_fraction", {}).get("mean"), d["config"].get("stage2_regrouped"), d["config"]["build_defines"], d.get("verify_all", {}).get("differing", d.get("verify_all", {}).get("error", "")))
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
S=25 bash scripts/packed_f32_repro/run.sh 2>&1 | grep -A2 "arm 5"
