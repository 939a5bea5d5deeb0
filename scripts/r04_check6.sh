# round 4, sixth GPU call: odd hops on the matrix-core path, AFC home columns, the bench line with traffic, 2.0 MS/s line, stage-2 kernels alone at the new signal
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c6; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wavefront_fft.py tests/test_golden.py -m gpu -q -k "other_formats or zero_copy or afc or wavefront_fft_variants or golden or end_to_end" > $O/parity.log 2>&1; tail -12 $O/parity.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 40"
timeout 300 python bench.py $N --sample-rate 2000000 2>$O/b2000.err | tail -n 1 > $O/bench_cfg3_2000k.json; cut -c1-300 $O/bench_cfg3_2000k.json; tail -2 $O/b2000.err
timeout 300 python bench.py $N --sample-rate 2400000 2>/dev/null | tail -n 1 > $O/bench_cfg3_2400k.json; cut -c1-200 $O/bench_cfg3_2400k.json
timeout 900 python bench.py --no-cpu-baseline 2>$O/bench_cfg3.err | tail -n 1 > $O/bench_cfg3.json; cut -c1-300 $O/bench_cfg3.json; tail -3 $O/bench_cfg3.err
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 8 --warmup 2"
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -- python bench.py $K > $O/kt_serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py $K > $O/kt.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*domain_stats.csv" -delete
