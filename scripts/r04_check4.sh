# round 4, fourth GPU call: the reference-side shim v2 (parts per GPU, GPU-served mixers, waterfall, file input), the mixer exchange entries, the default bench line
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c4; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_dropin_shim.py -m gpu -q > $O/shim.log 2>&1; tail -25 $O/shim.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "mixer or stage2 or opening_timer" > $O/parity.log 2>&1; tail -15 $O/parity.log
timeout 900 python bench.py 2>$O/bench_cfg3.err | tail -n 1 > $O/bench_cfg3.json; cut -c1-400 $O/bench_cfg3.json; tail -3 $O/bench_cfg3.err
timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40 --mixers 64 --force-dist 2>$O/mix.err | tail -n 1 > $O/bench_cfg3_mixers64.json; cut -c1-300 $O/bench_cfg3_mixers64.json; tail -3 $O/mix.err
