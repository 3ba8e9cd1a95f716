# A/B of serl_rollout_multi's workgroup placement (SERL_MIXED_PLACE, serl_mixed.h): bench lines and instruction-cache counters per mode
R=$(pwd); O=$R/gpurun_out/${1:-r05pl}; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o /tmp/hwid_probe 2>/dev/null && timeout 60 /tmp/hwid_probe > $O/hwid.json
for P in ${2:-0 1 2}; do
  (cd /tmp && export TMPDIR=/tmp && SERL_MIXED_PLACE=$P timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES -d $O/ic$P -o pi -- python $R/bench.py --workload mixed --fused --steps 1 --warmup 0 --no-cpu-baseline > $O/ic$P.log 2>&1)
  python tools/pmc_summary.py $O/ic$P | tr -d '\n' | sed "s/^/place $P /" | tee -a $O/icache.txt; echo | tee -a $O/icache.txt; rm -rf $O/ic$P
  SERL_MIXED_PLACE=$P timeout 200 python bench.py --workload mixed --fused --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('place $P 768 fused: %.4g M  %.2f ms' % (d['value']/1e6, d['ms_per_step']))" | tee -a $O/ab.txt
done
for P in ${3:-0 2}; do
  SERL_MIXED_PLACE=$P timeout 200 python bench.py --workload mixed --total-pop 2048 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('place $P 6144 fused: %.4g M  %.2f ms' % (d['value']/1e6, d['ms_per_step']))" | tee -a $O/ab.txt
done
