# The first GPU call of a round:   gpurun --timeout 1500 -- 'bash scripts/next_round_first.sh'
#  1. the whole GPU suite as the driver runs it (-x), and smoke();
#  2. the default bench line (what BENCH_rNN.json records), with its profiled children;
#  3. the three side paths that moved to the matrix cores in round 4, one line each (hops of 250 bytes, CF32 at both WAVE_RATEs).
# Everything lands under gpurun_out/first/; scripts/profile_round.sh + scripts/collect_profiles.py regenerate the whole of profiles/.
# Experiments round 4 left ready (each its own call): scripts/r05_masked_delay.sh (stage 2's delayed fetch only for lanes that use it: parity, time, counters of an experiment
# build, DESIGN 7.1 d) and scripts/r05_lds_layout_ab.sh (statistics on the wavefront FFT's rare event under twelve processes per GPU, profiles/r04_experiments.md I).
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/first; rm -rf $O; mkdir -p $O
(nproc; cat /sys/fs/cgroup/cpu.max) > $O/host_cpus.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $O/gpu_suite.log 2>&1; tail -14 $O/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout 900 python bench.py 2>$O/bench_cfg3.err | tail -n 1 > $O/bench_cfg3.json; cut -c1-400 $O/bench_cfg3.json; tail -2 $O/bench_cfg3.err
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4 --steps 24"
timeout 300 python bench.py $N --sample-rate 2000000 2>/dev/null | tail -n 1 > $O/bench_cfg3_2000k.json; cut -c1-200 $O/bench_cfg3_2000k.json
timeout 300 python bench.py $N --steps 12 --sample-format f32 --ring 1 --dongles 32768 2>/dev/null | tail -n 1 > $O/bench_f32_32768.json; cut -c1-200 $O/bench_f32_32768.json
timeout 300 python bench.py $N --steps 12 --workload cfg2 --sample-format f32 --ring 1 --dongles 16384 2>/dev/null | tail -n 1 > $O/bench_f32_am16384.json; cut -c1-200 $O/bench_f32_am16384.json
du -sh $O
