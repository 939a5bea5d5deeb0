# Round 5: are the MATRIX-CORE kernels right when processes share the GPU?  Three bench.py processes at once (4 096 dongles each, the int8 channelizer + stage 2: long launches),
# each checking 64 sampled dongles of its last batch against the oracle (decisions exact, audio <= 1e-4 RMS) -- twice.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/shared_bench; rm -rf $O; mkdir -p $O
python -c "import torch"
for round in 1 2; do
  pids=""
  for p in 1 2 3; do
    timeout 600 python bench.py --dongles 4096 --steps $((100 + 37 * p)) --warmup 2 --no-cpu-baseline --no-traffic --no-verify-all --verify 64 2>$O/err.$round.$p.txt | tail -n 1 > $O/bench.$round.$p.json &
    pids="$pids $!"
  done
  wait $pids
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench.*.json")):
    try:
        j = json.load(open(f)); print(f.split("/")[-1], "ms/step", j["ms_per_step"], "verified", j.get("verified_dongles"), str(j.get("verify"))[:200])
    except Exception as e:
        print(f, "unreadable", e)
PY
