# Round 6, call 7: window pieces software-pipelined across the workgroup barrier (fft_size >= 1024) -- parity, then A/B against the previous commit's library (_base)
# and three placements of the finish (AB_MID_AT 7 / 11 / 15; the product's is 13); matrix-pipe busy counter at fft 1024.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c7; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/rtlsdr-airband_amd
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "other_formats or stage1 or chunk or end_to_end" > $O/suite.log 2>&1; tail -3 $O/suite.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4"
for round in 1 2; do
  for l in new base mid7 mid11 mid15; do
    lib=$L/libairband_hip.so; [ $l = base ] && lib=$GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so
    case $l in mid*) lib=$L/libairband_hip_exp_$l.so;; esac
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 30 --fft-log 10 2>$O/err_${l}_fft1024_$round.log | tail -1 > $O/${l}_fft1024_$round.json
    AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 20 --fft-log 11 2>$O/err_${l}_fft2048_$round.log | tail -1 > $O/${l}_fft2048_$round.json
  done
done
for l in new base; do
  lib=$L/libairband_hip.so; [ $l = base ] && lib=$GRAFT_REPO_ROOT/_base/rtlsdr-airband_amd/libairband_hip.so
  AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 10 --fft-log 12 2>$O/err_${l}_fft4096.log | tail -1 > $O/${l}_fft4096.json
  AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 6 --fft-log 13 2>$O/err_${l}_fft8192.log | tail -1 > $O/${l}_fft8192.json
  AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 20 --fft-log 10 --sample-rate 2400000 2>$O/err_${l}_fft1024_2400k.log | tail -1 > $O/${l}_fft1024_2400k.json
  AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 20 --fft-log 10 --sample-format s16 2>$O/err_${l}_fft1024_cs16.log | tail -1 > $O/${l}_fft1024_cs16.json
  AIRBAND_HIP_LIB=$lib timeout 300 python bench.py $N --steps 30 2>$O/err_${l}_cfg3.log | tail -1 > $O/${l}_cfg3.json
  AIRBAND_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_$l -- python bench.py $N --verify 0 --steps 3 --warmup 1 --fft-log 10 > $O/pmc_mfma_$l.log 2>&1
  AIRBAND_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_fft1024_$l -- python bench.py $N --verify 0 --steps 8 --warmup 2 --fft-log 10 > $O/kt_fft1024_$l.log 2>&1
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c7"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, "verified", d.get("verified_dongles"), d["roofline"]["frac"], d["roofline"].get("mfma_frac"), d["config"]["build_defines"])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
PY
