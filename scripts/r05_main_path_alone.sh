# Round 5: four processes on one GPU, each running the MAIN path on replicated dongles (scripts/r05_exchange_stress.py, R05_MAIN_PATH=1), nothing else on the GPU.
# arg 1: seconds; arg 2: tag; arg 3: library ("" = product); arg 4: number of processes
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SECS=${1:-40}; TAG=${2:-product}; LIB=$3; P=${4:-4}
O=$GRAFT_REPO_ROOT/gpurun_out/main_path_alone_$TAG; rm -rf $O; mkdir -p $O
pids=""
for p in $(seq 1 $P); do AIRBAND_HIP_LIB=$LIB R05_MAIN_PATH=1 timeout $((SECS + 240)) python scripts/r05_exchange_stress.py $TAG $SECS u8 1024 $O/$TAG.jsonl > $O/$TAG.$p.log 2>&1 & pids="$pids $!"; done
wait $pids
grep -h EVENT $O/$TAG.*.log | cut -c1-330 | head -6
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$O/$TAG.jsonl")]
ch = collections.Counter()
first_events = 0
for r in rows:
    seen = set()
    for e in r.get("event_list", []):
        for c in e.get("channels", []): ch[c] += 1
        if (e["handle"]) not in seen:
            seen.add(e["handle"]); first_events += 1
print("$TAG: processes %d handles %d batches %d events (batches with a differing dongle) %d, handles with an event %d, channels seen differing %s" % (len(rows), sum(r["handles"] for r in rows), sum(r["batches"] for r in rows), sum(r["events"] for r in rows), first_events, dict(ch)))
PY
