# Round 6, call 8: CF32 on the float32 matrix pipe at fft 4096 / 8192 (window segments), with AFC (float tables re-tuned on the device), with hops of an odd number of
# samples; regrouping by residency (the 32 768-dongle shard); what the channelizer loses with fewer wavefronts per CU (AIRBAND_HIP_DFT_EXTRA_LDS).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c8; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "other_formats or afc or stage1 or chunk or cf32 or CF32 or lds" > $O/suite.log 2>&1; tail -3 $O/suite.log
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "residency or F32" > $O/scale.log 2>&1; tail -3 $O/scale.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4"
F="$N --sample-format f32 --ring 1 --dongles 32768"
for fl in 12 13; do
  timeout 300 python bench.py $F --steps 6 --fft-log $fl 2>$O/err_f32_fft$fl.log | tail -1 > $O/f32_fft$fl.json
  AIRBAND_BENCH_FLAGS=4 timeout 300 python bench.py $F --steps 4 --fft-log $fl 2>$O/err_f32_fft${fl}_force.log | tail -1 > $O/f32_fft${fl}_force_fft.json
done
timeout 300 python bench.py $F --steps 20 2>$O/err_f32.log | tail -1 > $O/f32_fft512.json
timeout 300 python bench.py $F --steps 20 --afc 2 2>$O/err_f32_afc.log | tail -1 > $O/f32_afc.json
AIRBAND_BENCH_FLAGS=4 timeout 300 python bench.py $F --steps 10 --afc 2 2>$O/err_f32_afc_force.log | tail -1 > $O/f32_afc_force_fft.json
timeout 300 python bench.py $F --steps 20 --sample-rate 2000000 2>$O/err_f32_2000k.log | tail -1 > $O/f32_2000k.json
AIRBAND_BENCH_FLAGS=4 timeout 300 python bench.py $F --steps 10 --sample-rate 2000000 2>$O/err_f32_2000k_force.log | tail -1 > $O/f32_2000k_force_fft.json
for r in 1 2; do
  for x in 0 2560 5632 14848; do
    AIRBAND_HIP_DFT_EXTRA_LDS=$x timeout 300 python bench.py $N --steps 30 2>/dev/null | tail -1 > $O/cfg3_extra${x}_$r.json
    AIRBAND_HIP_DFT_EXTRA_LDS=$x timeout 300 python bench.py $N --steps 30 --workload cfg2 --dongles 65536 2>/dev/null | tail -1 > $O/am65536_extra${x}_$r.json
  done
  timeout 300 python bench.py $N --steps 40 --workload cfg4 2>/dev/null | tail -1 > $O/cfg4_auto_$r.json
  timeout 300 python bench.py $N --steps 40 --workload cfg4 --regroup 0 2>/dev/null | tail -1 > $O/cfg4_rg0_$r.json
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c8"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["config"]["channelizer"], d["roofline"]["frac"], d["config"].get("stage2_regrouped"))
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
