# Round 5: the exchange kernel with and without s_waitcnt lgkmcnt(0) at every wavefront-level LDS exchange under a twelve-process load on one GPU (when the
# experiment ran, the wait was the experiment build -DAB_WAVE_SYNC_WAITS; it has shipped since, and the build without it is -DAB_WAVE_SYNC_NO_WAIT:
#   AIRBAND_EXTRA_DEFINES=-DAB_WAVE_SYNC_NO_WAIT AIRBAND_BUILD_TAG=sync_nowait python rtlsdr-airband_amd/_build.py),
# every process pushing replicated dongles through the exchange kernel and comparing them bit for bit (scripts/r05_exchange_stress.py).
#   gpurun --timeout 900 -- 'bash scripts/r05_exchange_stress.sh 50 12 small'      (shape: small = the fuzz campaign's handle sizes, big = hundreds of dongles)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SECS=${1:-50}; P=${2:-12}; SHAPE=${3:-small}
O=$GRAFT_REPO_ROOT/gpurun_out/exchange_stress_$SHAPE; rm -rf $O; mkdir -p $O
NOWAIT=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_sync_nowait.so
ls $NOWAIT || exit 1
python -c "import torch" # page the image in once, outside the timed arms
run_arm() { # name, processes, [library]
  local name=$1 procs=$2 lib=$3 pids=""
  for p in $(seq 1 $procs); do
    case $((p % 3)) in 0) fmt=u8;; 1) fmt=f32;; 2) fmt=f32x;; esac
    if [ $SHAPE = small ]; then n=$((1 + p % 5)); [ $fmt = f32x ] && n=$((1 + p % 3)); host=""; [ $((p % 2)) = 0 ] && host=host
    else n=1024; [ $fmt = f32 ] && n=192; [ $fmt = f32x ] && n=96; host=""; fi
    [ $n -lt 2 ] && n=2
    if [ -n "$lib" ]; then AIRBAND_HIP_LIB=$lib timeout $((SECS + 240)) python scripts/r05_exchange_stress.py $name $SECS $fmt $n $O/$name.jsonl $host > $O/$name.$p.log 2>&1 &
    else timeout $((SECS + 240)) python scripts/r05_exchange_stress.py $name $SECS $fmt $n $O/$name.jsonl $host > $O/$name.$p.log 2>&1 & fi
    pids="$pids $!"
  done
  wait $pids
  grep -h EVENT $O/$name.*.log | cut -c1-400 | head -20
  grep -L '"batches"' $O/$name.*.log | head -3 | while read f; do echo "== $f"; tail -5 $f; done
}
run_arm control $P $NOWAIT
run_arm waits $P ""
run_arm control2 $P $NOWAIT
python - <<PY
import json, glob, os
O = "$O"
print("| arm | processes | handles | batches (= launches of the exchange kernel) | hop transforms | events |")
print("|---|---|---|---|---|---|")
for f in sorted(glob.glob(O + "/*.jsonl")):
    rows = [json.loads(l) for l in open(f)]
    print("| %s | %d | %d | %d | %.3g | %d |" % (os.path.basename(f)[:-6], len(rows), sum(r["handles"] for r in rows), sum(r["batches"] for r in rows), sum(r["hop_transforms"] for r in rows), sum(r["events"] for r in rows)))
PY
