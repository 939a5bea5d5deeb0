# The first GPU call of the round after a CPU-only session (round 3 ended without GPU minutes; csrc/channelizer_fft.hip's exchange kernel has only
# run on the CPU emulation):   gpurun --timeout 900 -- 'bash scripts/next_round_first.sh'
#  1. parity of everything that runs on the wavefront-FFT path (f32, odd hops, FORCE_FFT, AFC's spectrum launch), then the rest of the quick subset;
#  2. its time: configs[2] forced onto it, f32 dongles, kernel trace;
#  3. two open questions of DESIGN 7.1 / 4.4 as counters: the instruction cache under the five stage-2 kernels side by side and alone, LDS bank
#     conflicts of the exchange kernel.  Counter names differ between rocprofv3 releases: the list the box offers is logged first.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/first; rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_wavefront_fft.py -m gpu -q -k "fft_wave64 or other_formats or afc or golden or wavefront_fft_variants" > $O/parity_fft.log 2>&1; tail -5 $O/parity_fft.log
# the same cases on the shuffle kernel: tells a fault of the exchange kernel from a fault of a case (2.0 MS/s and f32 at these sizes are new on the GPU either way)
AIRBAND_HIP_FFT_SHUFFLE=1 timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wavefront_fft.py -m gpu -q -k "fft_wave64 or SFMT_F32 or wavefront_fft_variants" > $O/parity_fft_shuffle.log 2>&1; tail -5 $O/parity_fft_shuffle.log
N="--no-cpu-baseline --no-traffic --no-verify-all --verify 4"
AIRBAND_BENCH_FLAGS=4 timeout 200 python bench.py $N --steps 6 --warmup 2 2>/dev/null | tail -n 1 > $O/bench_cfg3_force_fft.json; cut -c1-400 $O/bench_cfg3_force_fft.json
timeout 200 python bench.py $N --steps 6 --warmup 2 --sample-format f32 --ring 1 --dongles 32768 2>/dev/null | tail -n 1 > $O/bench_f32_32768.json; cut -c1-400 $O/bench_f32_32768.json
# the same two lines on the shuffle kernel the exchange kernel replaced (A/B on one box)
AIRBAND_HIP_FFT_SHUFFLE=1 AIRBAND_BENCH_FLAGS=4 timeout 300 python bench.py $N --steps 4 --warmup 1 2>/dev/null | tail -n 1 > $O/bench_cfg3_force_fft_shuffle.json; cut -c1-400 $O/bench_cfg3_force_fft_shuffle.json
AIRBAND_HIP_FFT_SHUFFLE=1 timeout 300 python bench.py $N --steps 4 --warmup 1 --sample-format f32 --ring 1 --dongles 32768 2>/dev/null | tail -n 1 > $O/bench_f32_32768_shuffle.json; cut -c1-400 $O/bench_f32_32768_shuffle.json
timeout 200 python bench.py $N --steps 40 2>/dev/null | tail -n 1 > $O/bench_cfg3.json; cut -c1-400 $O/bench_cfg3.json
K="--no-cpu-baseline --no-traffic --no-verify-all --verify 0 --steps 3 --warmup 1"
AIRBAND_BENCH_FLAGS=4 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_force_fft -- python bench.py $K > $O/kt_fft.log 2>&1
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|LDS_BANK|LDS_IDX|INSTS_LDS|SQ_INST_CYCLES_VMEM" | cut -c1-160 > $O/counters_offered.txt; head -40 $O/counters_offered.txt
timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_icache -- python bench.py $K > $O/pmc_icache.log 2>&1
AIRBAND_BENCH_FLAGS=8 timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_icache_serial -- python bench.py $K > $O/pmc_icache_serial.log 2>&1
AIRBAND_BENCH_FLAGS=4 timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc_lds_fft -- python bench.py $K > $O/pmc_lds_fft.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +4M -delete
du -sh $O
