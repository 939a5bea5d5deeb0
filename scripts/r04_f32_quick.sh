# round 4: the CF32 matrix-core kernel, quick loop -- parity of the f32 cases, then the two f32 bench lines
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/f32q; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "other_formats and SFMT_F32" > $O/parity.log 2>&1; tail -3 $O/parity.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 12 --warmup 2"
timeout 300 python bench.py $N --sample-format f32 --ring 1 --dongles 32768 2>$O/f32.err | tail -n 1 > $O/bench_f32_32768.json; tail -2 $O/f32.err
timeout 300 python bench.py $N --workload cfg2 --sample-format f32 --dongles 16384 --ring 1 2>$O/f32am.err | tail -n 1 > $O/bench_f32_am16384.json
python - <<PY
import json
for f in ["bench_f32_32768.json","bench_f32_am16384.json"]:
    d=json.load(open("$O/"+f)); r=d["roofline"]
    print(f, d["ms_per_step"], {k:round(v,2) for k,v in d["stage_ms"].items()}, r["kernel"], r["frac"], r.get("hbm_frac"), d.get("verified_dongles"))
PY
