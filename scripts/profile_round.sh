# Regenerates the round's measurements on a GPU box:  rm -rf gpurun_out/final; gpurun --timeout 1500 -- 'bash scripts/profile_round.sh'
# (gpurun MERGES into the local gpurun_out/, so remove the old copy first), then copy what is wanted into profiles/.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py 2>$O/bench_cfg3.err | tail -1 > $O/r01_bench_cfg3.json
timeout 300 python bench.py --no-cpu-baseline --workload cfg2 2>/dev/null | tail -1 > $O/r01_bench_cfg2.json
timeout 300 python bench.py --no-cpu-baseline --workload cfg2 --dongles 65536 2>/dev/null | tail -1 > $O/r01_bench_am65536.json
timeout 300 python bench.py --no-cpu-baseline --pipelined 2>/dev/null | tail -1 > $O/r01_bench_cfg3_pipelined.json
timeout 300 python bench.py --no-cpu-baseline --host-path 2>/dev/null | tail -1 > $O/r01_bench_cfg3_hostpath.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --no-cpu-baseline > $O/kt.log 2>&1
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -- python bench.py --no-cpu-baseline > $O/kt_serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH --output-format csv -d $O/pmc_sq -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_sq.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
cat $O/r01_bench_cfg3.json | cut -c1-400
du -sh $O
