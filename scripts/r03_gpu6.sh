# round 3, re-entry: state check of the committed build -- GPU suite (timed), the default bench line, kernel trace of configs[2]
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_6; rm -rf $O; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q -x --durations=15 ) > $O/pytest.log 2>&1; tail -25 $O/pytest.log
timeout 600 python bench.py --steps 40 2>$O/cfg3.err | tail -1 > $O/bench_cfg3.json; cut -c1-1500 $O/bench_cfg3.json
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 8 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3 -- python bench.py $K > $O/kt_cfg3.log 2>&1
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3_serial -- python bench.py $K > $O/kt_serial.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -14 $f | cut -c1-220; done
tail -5 $O/cfg3.err
