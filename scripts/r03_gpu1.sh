# round 3, GPU call 1: full GPU suite on the product library, then the tone-kernel unroll reproducer:
# the bit-exact stage-2 test, 4 times each, on libraries whose steady-state Goertzel loops are unrolled 25x (no SGPR spills) and 50x (38 SGPR spills)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_1; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for n in 25 50; do
  for i in 1 2 3 4; do
    AIRBAND_HIP_LIB=$GRAFT_REPO_ROOT/rtlsdr-airband_amd/libairband_hip_exp_tone$n.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "stage2_bit_exact or full_slot_blocks" > $O/tone${n}_$i.log 2>&1
    tail -1 $O/tone${n}_$i.log
  done
done
