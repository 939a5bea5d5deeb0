# A/B on ONE box: r02 library (tree at ca45e80 under _r02/), the round-3 channelizer without / with the in-wave tile pipeline
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_4; rm -rf $O; mkdir -p $O
N="--no-cpu-baseline --no-traffic --verify 0 --steps 60"
(while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done) > $O/clocks.log 2>&1 &
SMI=$!
for round in 1 2; do
  (cd _r02 && timeout 300 python bench.py $N 2>/dev/null | tail -1) > $O/A_r02_$round.json
  timeout 300 python bench.py $N 2>/dev/null | tail -1 > $O/C_pipe_$round.json
  AIRBAND_HIP_LIB=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_nopipe.so timeout 300 python bench.py $N 2>/dev/null | tail -1 > $O/B_nopipe_$round.json
done
(cd _r02 && timeout 300 python bench.py $N --workload cfg2 --dongles 65536 2>/dev/null | tail -1) > $O/A_r02_am.json
timeout 300 python bench.py $N --workload cfg2 --dongles 65536 2>/dev/null | tail -1 > $O/C_am.json
kill $SMI
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_4"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], d["stage_ms"]["channelizer"], d["stage_ms"]["demod"])
    except Exception as e: print(f, "ERR", e)
PY
grep -o '"sclk clock speed:": "[^"]*"' $O/clocks.log | sort | uniq -c | sort -rn | head -8
grep -o '"Current Socket Graphics Package Power (W)": "[^"]*"' $O/clocks.log | sort | uniq -c | sort -rn | head -5
head -c 600 $O/clocks.log
