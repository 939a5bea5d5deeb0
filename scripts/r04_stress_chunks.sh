set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/stress; rm -rf $O; mkdir -p $O
SEED=${1:-346}; ITERS=${2:-8}; PROCS=${3:-12}
for p in $(seq 1 $PROCS); do timeout 900 python scripts/r04_repro_chunks.py $SEED $ITERS p$p > $O/p$p.txt 2>&1 & done
wait
cat $O/p*.txt | grep -v amdgpu.ids | cut -c1-400 | head -80
