# round 4, third GPU call: f32 scale case, the PMC child run of bench.py with its stderr, open fraction against the signal's start batch, the box's CPU limits
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c3; rm -rf $O; mkdir -p $O
(nproc; cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us; python -c "import os; print(len(os.sched_getaffinity(0)))"; grep -c processor /proc/cpuinfo; cat /proc/loadavg) > $O/cpus.txt 2>&1; cat $O/cpus.txt
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "SFMT_F32" > $O/scale.log 2>&1; tail -4 $O/scale.log
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_dropin_shim.py -m gpu -x -q -k "stage2 or opening_timer or reference_harness" > $O/parity.log 2>&1; tail -4 $O/parity.log
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_child -- python bench.py --child --no-verify-all --workload cfg3 --steps 3 --warmup 1 --ring 3 --signal-start-batch 1 > $O/pmc_child.log 2>&1; tail -5 $O/pmc_child.log; find $O/pmc_child -name "*.csv" | head
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 24"
for S in 2 3 4 5; do timeout 300 python bench.py $N --signal-start-batch $S 2>/dev/null | tail -n 1 > $O/bench_start$S.json; python -c "
import json; d=json.load(open('$O/bench_start$S.json')); print($S, d['ms_per_step'], d['stage_ms'], d.get('open_fraction'), d.get('verified_dongles'))"; done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +4M -delete
