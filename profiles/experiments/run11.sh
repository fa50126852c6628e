mkdir -p gpurun_out/exp
timeout 900 python -m pytest tests/test_node_jets.py tests/test_gpu_parity.py -q -m gpu --timeout=300 -k "reference_literal or contraction_off or outer_ss or two_body" > gpurun_out/exp/t11.log 2>&1; tail -15 gpurun_out/exp/t11.log | cut -c1-300
