# the default bench line alone (what the driver runs at round end), into gpurun_out/final for scripts/collect_profiles.py
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
timeout 900 python bench.py 2>$O/bench_cfg3.err | tail -n 1 > $O/r04_bench_cfg3.json; cut -c1-300 $O/r04_bench_cfg3.json; tail -3 $O/bench_cfg3.err
