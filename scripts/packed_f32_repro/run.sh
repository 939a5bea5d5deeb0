# gpurun --timeout 600 -- 'bash scripts/packed_f32_repro/run.sh'      (builds the reproducer on the box; S = seconds per arm)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/packed_f32_repro; rm -rf $O; mkdir -p $O
S=${S:-40}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/packed_f32_repro/repro.hip -o $O/repro || exit 1
echo "arm 1: victim alone (packed)"; $O/repro victim $S | tee $O/arm1_victim_alone.txt
echo "arm 2: packed victim beside three aggressor processes"
for i in 1 2 3; do $O/repro aggressor $((S + 6)) > $O/arm2_aggressor$i.txt & done
sleep 3; $O/repro victim $S | tee $O/arm2_victim_with_aggressors.txt; wait
echo "arm 3: scalar victim (no packed instruction) beside three aggressor processes"
for i in 1 2 3; do $O/repro aggressor $((S + 6)) > $O/arm3_aggressor$i.txt & done
sleep 3; $O/repro victim_scalar $S | tee $O/arm3_scalar_victim_with_aggressors.txt; wait
echo "arm 4: two packed victims beside each other and three aggressors"
for i in 1 2 3; do $O/repro aggressor $((S + 6)) > $O/arm4_aggressor$i.txt & done
sleep 3; $O/repro victim $S > $O/arm4_victim_b.txt & $O/repro victim $S | tee $O/arm4_victim_a.txt; wait; cat $O/arm4_victim_b.txt
echo "arm 5: packed victim beside three PRODUCT aggressors (bench.py --dongles 4096 loops: the load under which libairband_hip's own kernels failed in round 5)"
for i in 1 2 3; do (while [ ! -f $O/stop ]; do timeout 120 python bench.py --dongles 4096 --steps 1500 --warmup 2 --no-cpu-baseline --no-traffic --no-verify-all --verify 0 > /dev/null 2>&1; done) & done
sleep 25; $O/repro victim $S | tee $O/arm5_victim_with_product_aggressors.txt; touch $O/stop; wait
rm -f $O/repro $O/stop
