# Regenerates the round's measurements on a GPU box:  rm -rf gpurun_out/final; gpurun --timeout 2400 -- 'bash scripts/profile_round.sh'
# (gpurun MERGES into the local gpurun_out/, so remove the old copy first), then `python scripts/collect_profiles.py r06` copies the summaries into profiles/.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${R:-r06}
O=gpurun_out/final; rm -rf $O; mkdir -p $O
# LIMIT=<seconds>: commands that would start after that many seconds of this script are skipped (most important first; a skipped measurement keeps the file of the
# previous run of the round in profiles/ -- scripts/collect_profiles.py only copies what exists)
T0=$(date +%s); LIMIT=${LIMIT:-100000}
ok() { [ $(( $(date +%s) - T0 )) -lt $LIMIT ] || { echo "skipped (time)"; return 1; }; }
nproc > $O/host.txt; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
ok && timeout 900 python bench.py 2>$O/bench_cfg3.err | tail -n 1 > $O/${R}_bench_cfg3.json; cut -c1-300 $O/${R}_bench_cfg3.json
N="--no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 4 --steps 40"
K="--no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 0 --steps 8 --warmup 2"
ok && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3 -- python bench.py $K > $O/kt_cfg3.log 2>&1
ok && AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3_serial -- python bench.py $K > $O/kt_serial.log 2>&1
ok && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 0 --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
ok && AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH --output-format csv -d $O/pmc_sq_serial -- python bench.py --no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 0 --steps 2 --warmup 1 > $O/pmc_sq.log 2>&1
ok && timeout 300 python bench.py $N --workload cfg2 --steps 400 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg2.json
ok && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg2 -- python bench.py $K --workload cfg2 > $O/kt_cfg2.log 2>&1
ok && AIRBAND_BENCH_FLAGS=4 timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 4 --steps 6 --warmup 2 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_force_fft.json
ok && timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 4 --steps 12 --warmup 2 --sample-format f32 --ring 1 --dongles 32768 2>/dev/null | tail -n 1 > $O/${R}_bench_f32_32768.json
ok && AIRBAND_BENCH_FLAGS=4 timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 4 --steps 4 --warmup 2 --sample-format f32 --ring 1 --dongles 32768 2>/dev/null | tail -n 1 > $O/${R}_bench_f32_32768_force_fft.json
ok && timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 4 --steps 6 --warmup 2 --sample-format f32 --ring 1 --dongles 32768 --fft-log 12 2>/dev/null | tail -n 1 > $O/${R}_bench_f32_32768_fft4096.json
ok && timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 4 --steps 12 --warmup 2 --sample-format f32 --ring 1 --dongles 32768 --afc 2 2>/dev/null | tail -n 1 > $O/${R}_bench_f32_32768_afc.json
ok && timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 4 --steps 12 --warmup 2 --sample-format f32 --ring 1 --dongles 32768 --sample-rate 2000000 2>/dev/null | tail -n 1 > $O/${R}_bench_f32_32768_2000k.json
ok && timeout 300 python bench.py $N --fft-log 12 --steps 12 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_fft4096.json
ok && timeout 300 python bench.py $N --fft-log 13 --steps 8 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_fft8192.json
ok && timeout 300 python bench.py $N --workload cfg4 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg4_shard.json
ok && timeout 300 python bench.py $N --workload cfg4 --regroup 0 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg4_shard_slot_order.json
ok && timeout 300 python bench.py $N --distinct-plans 65536 --verify 16 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_plans65536.json
ok && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg4 -- python bench.py $K --workload cfg4 > $O/kt_cfg4.log 2>&1
ok && timeout 300 python bench.py $N --afc 2 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_afc.json
ok && AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3_afc -- python bench.py $K --afc 2 > $O/kt_afc.log 2>&1
ok && AIRBAND_BENCH_FLAGS=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3_force_fft -- python bench.py --no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 0 --steps 3 --warmup 1 > $O/kt_fft.log 2>&1
ok && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_f32 -- python bench.py $K --sample-format f32 --ring 1 --dongles 32768 > $O/kt_f32.log 2>&1
ok && timeout 300 python bench.py $N --sample-format s16 --ring 1 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_cs16.json
ok && timeout 300 python bench.py $N --sample-rate 2000000 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_2000k.json
ok && timeout 300 python bench.py $N --fft-log 10 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_fft1024.json
ok && timeout 300 python bench.py $N --fft-log 11 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_fft2048.json
ok && timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 4 --steps 12 --warmup 2 --workload cfg2 --sample-format f32 --ring 1 --dongles 16384 2>/dev/null | tail -n 1 > $O/${R}_bench_f32_am16384.json
ok && timeout 300 python bench.py $N --workload cfg2 --dongles 65536 2>/dev/null | tail -n 1 > $O/${R}_bench_am65536.json
ok && timeout 300 python bench.py $N --mixers 64 --force-dist 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_mixers64.json
ok && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --no-cpu-baseline --no-traffic --no-verify-all --no-throughput-mode --verify 0 --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
ok && timeout 300 python bench.py $N --workload cfg2 --steps 400 --pipelined 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg2_pipelined.json
ok && timeout 300 python bench.py $N --pipelined 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_pipelined.json
ok && timeout 300 python bench.py $N --host-path --host-threads 32 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_hostpath.json
ok && timeout 300 python bench.py $N --sample-rate 2400000 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_2400k.json
ok && timeout 300 python bench.py $N --fft-log 8 2>/dev/null | tail -n 1 > $O/${R}_bench_cfg3_fft256.json
ok && AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg4_serial -- python bench.py $K --workload cfg4 > $O/kt_cfg4_serial.log 2>&1
ok && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_am65536 -- python bench.py $K --workload cfg2 --dongles 65536 > $O/kt_am.log 2>&1
ok && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cs16 -- python bench.py $K --sample-format s16 --ring 1 > $O/kt_cs16.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
du -sh $O
