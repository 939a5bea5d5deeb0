# Is the rare difference test_results_do_not_depend_on_how_the_bytes_arrive saw (only CF32 at fft >= 4096: workgroups with > 64 KiB of LDS, only with 12 processes on the GPU) tied to
# other processes' long kernels sharing the GPU?  Victims: the two seeds, N iterations each, in several processes.  Aggressors: bench.py loops with heavy kernels (arg 3 = how many).
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/preempt; rm -rf $O; mkdir -p $O
ITERS=${1:-25}; VICTIMS=${2:-6}; AGGR=${3:-4}
for a in $(seq 1 $AGGR); do
  (for k in 1 2 3 4 5 6; do timeout 300 python bench.py --dongles 2048 --fft-log 13 --steps 60 --warmup 2 --no-cpu-baseline --no-traffic --no-verify-all --verify 0 > $O/aggr$a.$k.txt 2>&1; done) &
done
sleep 20
for p in $(seq 1 $VICTIMS); do
  s=346; [ $((p % 2)) = 0 ] && s=32
  timeout 600 python scripts/r04_repro_chunks.py $s $ITERS v$p > $O/v$p.txt 2>&1 &
  VP="$VP $!"
done
wait $VP
cat $O/v*.txt | grep -v amdgpu.ids | cut -c1-600 | head -60
kill %1 %2 %3 %4 2>/dev/null; sleep 1
tail -n 1 $O/aggr1.1.txt | cut -c1-200
