for f in "" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1" "-mllvm -enable-post-misched=0" "-mllvm -amdgpu-schedule-metric-bias=0" "-mllvm -amdgpu-disable-rewrite-mfma-form-sched-stage=1" "-mllvm -misched-cluster=0"; do
  echo "FLAGS=[$f]"
  HEYOKA_AMD_HIPRTC_FLAGS="$f" python bench.py --workload outer_ss --steps 6 --warmup 2 --no-cpu-baseline --no-extra-workloads 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; print('  value %.4g frac %.4f kernel_ms %.2f spills %s'%(d['value'], r['frac'], r['kernel_ms_avg'], r['kernel_resources']['vgpr_spill']))
except Exception as e: print('  ERR', e)
"
done
