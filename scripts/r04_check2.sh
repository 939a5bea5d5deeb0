# round 4, second GPU call: the batch-boundary fix of the lowpass kind, the FFT-path scale cases, and the default bench line with the new CPU baseline / open_fraction
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c2; rm -rf $O; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "opening_timer or stage2 or end_to_end or full_slot" > $O/parity.log 2>&1; tail -4 $O/parity.log
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "fft_wave64 or SFMT_F32 or 4096_mixed_splits2" > $O/scale.log 2>&1; tail -4 $O/scale.log
timeout 900 python bench.py 2>$O/bench_cfg3.err | tail -n 1 > $O/bench_cfg3.json; cut -c1-600 $O/bench_cfg3.json
timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40 --signal-start-batch 0 2>/dev/null | tail -n 1 > $O/bench_cfg3_start0.json; cut -c1-300 $O/bench_cfg3_start0.json
