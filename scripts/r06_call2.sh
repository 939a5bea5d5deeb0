# Round 6, call 2: (1) the cache-policy matrix of the channelizer's read-once LDS-DMA pieces; (2) first run of fleets with distinct channel plans (device-built tables).
#   bash /tmp/build_variants.sh  (AIRBAND_EXTRA_DEFINES=-DAB_DMA_NT_AUX=<aux> [-DAB_DMA_FIRST_AUX=2], tags nt nt0 ntsc0 sc1 ntsc1 sc0sc1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c2; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
for round in 1 2; do
  for l in base nt nt0 ntsc0 sc1 ntsc1 sc0sc1; do
    lib=$L/libairband_hip.so; [ $l != base ] && lib=$L/libairband_hip_exp_$l.so
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N 2>$O/err_${l}_$round.log | tail -1 > $O/${l}_cfg3_$round.json
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --workload cfg2 --dongles 65536 2>/dev/null | tail -1 > $O/${l}_am65536_$round.json
  done
done
for p in 64 4096 65536; do
  timeout 600 python bench.py --no-cpu-baseline --no-traffic --verify 16 --steps 40 --distinct-plans $p 2>$O/err_plans$p.log | tail -1 > $O/plans${p}.json
done
AIRBAND_HIP_HOST_TABLES_MAX=2 timeout 600 python bench.py --no-cpu-baseline --no-traffic --verify 16 --steps 10 --distinct-plans 64 --dongles 4096 2>$O/err_plans64_dev.log | tail -1 > $O/plans64_device_built.json
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c2"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["config"]["channelizer"], d["config"]["build_defines"], d.get("verify", {}).get("error", ""))
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
tail -5 $O/err_plans65536.log
