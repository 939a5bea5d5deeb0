# One short box: the whole GPU suite on the library under test, then new / base / new / base bench lines of the default workload (configs[2]).
#   (build _base as scripts/ab_bench.sh says)   gpurun --timeout 240 -- 'bash scripts/ab_quick.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/abq; rm -rf $O; mkdir -p $O
timeout 135 python -m pytest tests -m gpu -x -q > $O/parity.log 2>&1; tail -3 $O/parity.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 30 --warmup 3"
i=0
for l in new base new base; do
  i=$((i+1))
  lib=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip.so; [ $l = base ] && lib=$GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so
  AIRBAND_HIP_LIB=$lib timeout 40 python bench.py $N 2>/dev/null | tail -1 > $O/${l}_$i.json
  python - $O/${l}_$i.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
