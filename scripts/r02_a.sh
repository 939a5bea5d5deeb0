# round 2, GPU call A: full GPU test suite (incl. scale parity), default bench line, first variants, kernel traces the verdict asked for
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; rm -rf $O; mkdir -p $O
nproc > $O/nproc.txt; grep -m1 "model name" /proc/cpuinfo >> $O/nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; tail -30 $O/pytest.log
timeout 600 python bench.py --steps 40 --warmup 5 2>$O/bench_cfg3.err | tail -1 > $O/bench_cfg3.json; cut -c1-1500 $O/bench_cfg3.json
AIRBAND_HIP_LIB=$PWD/rtlsdr-airband_amd/libairband_hip_exp_ntstore.so timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-traffic --verify 4 2>/dev/null | tail -1 > $O/bench_ntstore.json; cut -c1-300 $O/bench_ntstore.json
timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-traffic --verify 0 --workload cfg2 2>/dev/null | tail -1 > $O/bench_cfg2.json
timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-traffic --verify 0 --workload cfg2 --dongles 65536 2>/dev/null | tail -1 > $O/bench_am65536.json
AIRBAND_BENCH_FLAGS=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_fft -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 3 --warmup 1 > $O/kt_fft.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3 -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 8 --warmup 2 > $O/kt_cfg3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg2 -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 8 --warmup 2 --workload cfg2 > $O/kt_cfg2.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
du -sh $O
