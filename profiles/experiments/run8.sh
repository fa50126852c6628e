mkdir -p gpurun_out/exp
for t in np1body8_default np1body13_default np1body8_masses; do
timeout 90 python -m pytest tests/test_models.py -x -q -m gpu -k "test_models_step_and_propagate_vs_oracle and $t" > gpurun_out/exp/t8_$t.log 2>&1; echo "$t rc=$?"; tail -3 gpurun_out/exp/t8_$t.log | cut -c1-300
done
HEYOKA_AMD_PAIR_SPLIT=0 timeout 90 python -m pytest tests/test_models.py -x -q -m gpu -k "test_models_step_and_propagate_vs_oracle and np1body8_default" > gpurun_out/exp/t8_ctl.log 2>&1; echo "ctl rc=$?"; tail -3 gpurun_out/exp/t8_ctl.log | cut -c1-300
python - <<'PY'
import sys; sys.path.insert(0,'.')
import heyoka_amd as hy
ta = hy.taylor_adaptive_batch(hy.model.np1body(8), None, 64)
print(ta.hip_source_mode)
PY
