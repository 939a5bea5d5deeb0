set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; rm -rf $O; mkdir -p $O
export AIRBAND_HIP_LIB=$PWD/rtlsdr-airband_amd/libairband_hip$V.so
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 6 --warmup 2 > $O/kt_serial.log 2>&1
cat $O/kt_serial/*/*kernel_stats.csv | head -8 | cut -c1-150
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH --output-format csv -d $O/pmc_sq -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 2 --warmup 1 > $O/pmc_sq.log 2>&1
AIRBAND_BENCH_FLAGS=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $O/pmc_sq2 -- python bench.py --no-cpu-baseline --no-traffic --verify 0 --steps 2 --warmup 1 > $O/pmc_sq2.log 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
unset AIRBAND_HIP_LIB
python - <<'PY'
import torch, importlib, sys, time
sys.path.insert(0,'.')
pkg=importlib.import_module('rtlsdr-airband_amd')
def free(): 
    torch.cuda.synchronize(); return round(torch.cuda.mem_get_info()[0]/2**30,1)
print('start free', free())
chans,car=pkg.siggen.baseline_plan(mixed=True)
devs=[dict(channels=chans) for _ in range(65536)]
hip=pkg.AirbandHip(devs, wave_rate=16000, flags=1)
print('after handle', free())
iq=torch.empty((65536, 2624000), dtype=torch.uint8, device='cuda')
print('after iq', free())
hip.set_signal_plan(car); hip.generate_iq(iq.data_ptr(), 2624000, 0, 2600000); hip.synchronize()
for i in range(2): hip.process_device(iq.data_ptr(), 2624000)
hip.synchronize()
r=hip.collect(first_channel=0,n_channels=8,stats=True); t=hip.read_trace(0,8)
print('after run', free())
hip.close(); print('after close', free())
del iq; torch.cuda.empty_cache(); print('after del iq', free())
time.sleep(3); print('3s later', free())
PY
