mkdir -p gpurun_out/exp
timeout 1500 python -m pytest tests -q -m gpu --timeout=200 -x > gpurun_out/exp/t9.log 2>&1; tail -8 gpurun_out/exp/t9.log
