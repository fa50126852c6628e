mkdir -p gpurun_out/exp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_cpp_api.py -q -m gpu --timeout=300 -k "reference_batch_semantics or compact_mode or include_layout or cpp_api or nonfinite or tutorial" > gpurun_out/exp/t12.log 2>&1; tail -15 gpurun_out/exp/t12.log | cut -c1-400
