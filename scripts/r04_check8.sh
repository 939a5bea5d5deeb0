# round 4: CF32 on the float32 matrix pipe -- parity, scale, bench; the default bench line's traffic children
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c8; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "other_formats" > $O/parity.log 2>&1; tail -12 $O/parity.log
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -k "SFMT_F32" > $O/scale.log 2>&1; tail -6 $O/scale.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 12 --warmup 2"
timeout 300 python bench.py $N --sample-format f32 --ring 1 --dongles 32768 2>$O/f32.err | tail -n 1 > $O/bench_f32_32768.json; cut -c1-300 $O/bench_f32_32768.json; tail -2 $O/f32.err
AIRBAND_BENCH_FLAGS=4 timeout 300 python bench.py $N --steps 4 --sample-format f32 --ring 1 --dongles 32768 2>/dev/null | tail -n 1 > $O/bench_f32_32768_fft.json; cut -c1-200 $O/bench_f32_32768_fft.json
timeout 300 python bench.py $N --workload cfg2 --sample-format f32 --dongles 16384 --ring 1 2>$O/f32am.err | tail -n 1 > $O/bench_f32_am16384.json; cut -c1-300 $O/bench_f32_am16384.json; tail -2 $O/f32am.err
timeout 900 python bench.py --no-cpu-baseline 2>$O/bench_cfg3.err | tail -n 1 > $O/bench_cfg3.json; cut -c1-200 $O/bench_cfg3.json; tail -3 $O/bench_cfg3.err
