set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_5; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
N="--no-cpu-baseline --no-traffic --verify 8 --steps 30"
timeout 300 python bench.py $N 2>$O/cfg3.err | tail -1 > $O/bench_cfg3.json
timeout 300 python bench.py $N --afc 2 2>$O/afc.err | tail -1 > $O/bench_cfg3_afc.json
timeout 300 python bench.py $N --fft-log 12 --steps 10 2>$O/f12.err | tail -1 > $O/bench_fft4096.json
timeout 300 python bench.py $N --fft-log 13 --steps 6 2>$O/f13.err | tail -1 > $O/bench_fft8192.json
timeout 300 python bench.py $N --fft-log 10 2>/dev/null | tail -1 > $O/bench_fft1024.json
timeout 300 python bench.py $N --sample-format s16 --ring 1 2>/dev/null | tail -1 > $O/bench_cs16.json
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r03_5/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(os.path.basename(f), d["ms_per_step"], "chan %.3f demod %.3f"%(d["stage_ms"]["channelizer"], d["stage_ms"]["demod"]), r["bound"], r["frac"], "e2e", r.get("end_to_end_frac"), "mfma", r.get("mfma_frac"), "verified", d.get("verified_dongles"), d["config"]["channelizer"])
    except Exception as e: print(f, "ERR", e, open(f).read()[:300])
PY
tail -3 $O/*.err
