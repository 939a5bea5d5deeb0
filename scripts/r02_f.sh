set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for T in 1 4 16 32; do
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-traffic --verify 0 --host-path --host-threads $T --host-dongles 2048 2>$O/hp$T.err | tail -1 > $O/bench_hp$T.json
python -c "import json; j=json.load(open('$O/bench_hp$T.json')); print('HOSTPATH', $T, j.get('host_path'))"
done
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-traffic --verify 0 --host-path --host-threads 32 --host-dongles 8192 2>/dev/null | tail -1 > $O/bench_hp_8k.json
python -c "import json; j=json.load(open('$O/bench_hp_8k.json')); print('HOSTPATH 8192', j.get('host_path'))"
