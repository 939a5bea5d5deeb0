# Round 6, call 4: which kinds pay for regrouping (AIRBAND_HIP_REGROUP_KINDS), with and without the masked delayed fetch; nt policy, more samples; the packed-f32 reproducer.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c4; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
for round in 1 2 3; do
  for l in base md; do
    lib=$L/libairband_hip.so; [ $l != base ] && lib=$L/libairband_hip_exp_$l.so
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --regroup 0 2>/dev/null | tail -1 > $O/${l}_off_$round.json
    for k in 0x0C 0x04 0x08; do
      AIRBAND_HIP_REGROUP_KINDS=$k AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --regroup 1 2>/dev/null | tail -1 > $O/${l}_kinds${k}_$round.json
    done
  done
  AIRBAND_HIP_LIB=$L/libairband_hip_exp_nt.so timeout 300 python bench.py $N 2>/dev/null | tail -1 > $O/nt_off_$round.json
done
for k in 0x0C; do
  AIRBAND_HIP_REGROUP_KINDS=$k timeout 300 python bench.py $N --regroup 1 --key-on-s 0.15 2>/dev/null | tail -1 > $O/base_kinds${k}_duty10.json
  AIRBAND_HIP_REGROUP_KINDS=$k AIRBAND_HIP_LIB=$L/libairband_hip_exp_md.so timeout 300 python bench.py $N --regroup 1 --key-on-s 0.15 2>/dev/null | tail -1 > $O/md_kinds${k}_duty10.json
done
timeout 300 python bench.py $N --regroup 0 --key-on-s 0.15 2>/dev/null | tail -1 > $O/base_off_duty10.json
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c4"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), "open", d.get("open_fraction", {}).get("mean"), d["config"].get("stage2_regrouped"), d["config"]["build_defines"])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
S=25 bash scripts/packed_f32_repro/run.sh 2>&1 | tail -20
