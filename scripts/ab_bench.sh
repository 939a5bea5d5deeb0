# A/B harness, the common shape of the round-3 experiments (profiles/r03_experiments.md): on ONE gpurun box, first the parity subset on the library under
# test, then interleaved bench lines of the committed build against the variants.
#   git archive HEAD rtlsdr-airband_amd include | tar -x -C _base && (cd _base && python -c "import importlib; importlib.import_module('rtlsdr-airband_amd')._build.build()")
#   AIRBAND_EXTRA_DEFINES="-DAB_X" AIRBAND_BUILD_TAG=x python -c "import importlib; importlib.import_module('rtlsdr-airband_amd')._build.build()"   # variants, optional
#   gpurun --timeout 1500 -- 'bash scripts/ab_bench.sh "tests/test_gpu_parity.py tests/test_golden.py" "" "--workload cfg2" -- x'
# arguments: <pytest targets> <bench flags of workload 1> [<bench flags of workload 2> ...] -- [<variant tags> ...]
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ab; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
TESTS=$1; shift
WORK=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do WORK+=("$1"); shift; done; [ "$1" = "--" ] && shift
LIBS=("base:$GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so" "new:$L/libairband_hip.so")
for t in "$@"; do LIBS+=("$t:$L/libairband_hip_exp_$t.so"); done
[ -n "$TESTS" ] && { timeout 1200 python -m pytest $TESTS -m gpu -x -q > $O/parity.log 2>&1; tail -3 $O/parity.log; }
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
for round in 1 2; do
  for w in "${!WORK[@]}"; do
    for l in "${LIBS[@]}"; do
      AIRBAND_HIP_LIB=${l#*:} timeout 300 python bench.py $N ${WORK[$w]} 2>/dev/null | tail -1 > $O/${l%%:*}_w${w}_$round.json
    done
  done
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/ab"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"))
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
