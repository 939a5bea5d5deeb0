# Round 5: scripts/r05_repeat_stress.py under P processes: arm `churn` (random configurations per handle), arm `same` (one configuration per worker), arm `solo` (ONE process, churn).
#   gpurun --timeout 900 -- 'bash scripts/r05_repeat_stress.sh 60 12'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SECS=${1:-60}; P=${2:-12}
O=$GRAFT_REPO_ROOT/gpurun_out/repeat_stress; rm -rf $O; mkdir -p $O
python -c "import torch"
run_arm() { # name, processes, mode
  local name=$1 procs=$2 mode=$3 pids=""
  for p in $(seq 1 $procs); do
    timeout $((SECS + 240)) python scripts/r05_repeat_stress.py $name $SECS $((1000 + p)) $O/$name.jsonl $mode > $O/$name.$p.log 2>&1 &
    pids="$pids $!"
  done
  wait $pids
  grep -h EVENT $O/$name.*.log | cut -c1-700 | head -12
  grep -L '"launches"' $O/$name.*.log | head -3 | while read f; do echo "== $f"; tail -5 $f; done
}
run_arm churn $P ""
run_arm same $P same
run_arm solo 1 ""
python - <<PY
import json, glob, os
O = "$O"
print("| arm | processes | handles | launches | hop transforms | events |")
print("|---|---|---|---|---|---|")
for f in sorted(glob.glob(O + "/*.jsonl")):
    rows = [json.loads(l) for l in open(f)]
    print("| %s | %d | %d | %d | %.3g | %d |" % (os.path.basename(f)[:-6], len(rows), sum(r["handles"] for r in rows), sum(r["launches"] for r in rows), sum(r["hop_transforms"] for r in rows), sum(r["events"] for r in rows)))
PY
