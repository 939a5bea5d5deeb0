set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_3; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
N="--no-cpu-baseline --no-traffic --verify 8 --steps 40"
timeout 300 python bench.py $N 2>$O/cfg3.err | tail -1 > $O/bench_cfg3.json; cut -c1-400 $O/bench_cfg3.json
timeout 300 python bench.py $N --workload cfg2 --dongles 65536 2>/dev/null | tail -1 > $O/bench_am65536.json; cut -c1-300 $O/bench_am65536.json
timeout 300 python bench.py $N --workload cfg2 2>/dev/null | tail -1 > $O/bench_cfg2.json; cut -c1-300 $O/bench_cfg2.json
timeout 300 python bench.py $N --sample-rate 2400000 2>/dev/null | tail -1 > $O/bench_2400k.json; cut -c1-300 $O/bench_2400k.json
timeout 300 python bench.py $N --sample-format s16 --ring 1 2>/dev/null | tail -1 > $O/bench_cs16.json; cut -c1-300 $O/bench_cs16.json
timeout 300 python bench.py $N --fft-log 10 2>/dev/null | tail -1 > $O/bench_fft1024.json; cut -c1-300 $O/bench_fft1024.json
K="--no-cpu-baseline --no-traffic --verify 0 --steps 8 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3 -- python bench.py $K > $O/kt_cfg3.log 2>&1
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3_serial -- python bench.py $K > $O/kt_serial.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -12 $f | cut -c1-200; done
