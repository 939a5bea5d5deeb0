# Round 6, call 9: the pipelined mode (stage 1 of batch k beside stage 2 of batch k - 1) with the channelizer held to fewer wavefronts per CU (AIRBAND_HIP_DFT_EXTRA_LDS:
# 7 / 6 / 4 per CU cost it 0 / 1 / 12 % when it runs alone, call 8), i.e. with room on every CU for stage-2 wavefronts; distinct channel plans with PMC traffic.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c9; rm -rf $O; mkdir -p $O
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4"
for r in 1 2; do
  timeout 300 python bench.py $N --steps 40 2>/dev/null | tail -1 > $O/cfg3_seq_$r.json
  for x in 0 2560 5632 9000 14848; do
    AIRBAND_HIP_DFT_EXTRA_LDS=$x timeout 300 python bench.py $N --steps 40 --pipelined 2>$O/err_pipe_$x.log | tail -1 > $O/cfg3_pipelined_extra${x}_$r.json
  done
  for x in 0 5632 14848; do
    AIRBAND_HIP_DFT_EXTRA_LDS=$x timeout 300 python bench.py $N --steps 40 --pipelined --workload cfg2 --dongles 65536 2>/dev/null | tail -1 > $O/am65536_pipelined_extra${x}_$r.json
    AIRBAND_HIP_DFT_EXTRA_LDS=$x timeout 300 python bench.py $N --steps 40 --pipelined --workload cfg4 2>/dev/null | tail -1 > $O/cfg4_pipelined_extra${x}_$r.json
  done
done
timeout 600 python bench.py --no-cpu-baseline --no-verify-all --verify 16 --steps 30 --distinct-plans 65536 2>$O/err_plans65536.log | tail -1 > $O/plans65536_traffic.json
timeout 600 python bench.py --no-cpu-baseline --no-verify-all --verify 16 --steps 30 --distinct-plans 4096 2>$O/err_plans4096.log | tail -1 > $O/plans4096_traffic.json
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c9"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["config"]["schedule"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["config"].get("stage2_regrouped"))
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
