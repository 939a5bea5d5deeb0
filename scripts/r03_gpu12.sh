# round 3: full GPU suite on the final kernels + the AFC line after the "re-tune only when something moved" change
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_12; rm -rf $O; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
timeout 300 python bench.py $N 2>/dev/null | tail -1 > $O/bench_cfg3.json
timeout 300 python bench.py $N --afc 2 2>/dev/null | tail -1 > $O/bench_cfg3_afc.json
timeout 300 python bench.py $N 2>/dev/null | tail -1 > $O/bench_cfg3_b.json
timeout 300 python bench.py $N --afc 2 2>/dev/null | tail -1 > $O/bench_cfg3_afc_b.json
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 8 --warmup 2"
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3_afc -- python bench.py $K --afc 2 > $O/kt_afc.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -14 $f | cut -c1-200; done
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_12"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["ms_per_step"], "chan %.3f demod %.3f"%(d["stage_ms"]["channelizer"], d["stage_ms"]["demod"]), "verified", d.get("verified_dongles"))
    except Exception as e: print(f, "ERR", e)
PY
