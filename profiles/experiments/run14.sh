mkdir -p gpurun_out/exp
timeout 900 python -m pytest tests/test_models.py tests/test_gpu_parity.py -q -m gpu --timeout=240 -x -k "models or parameter_masses or wave_role or random_systems or block_mode or nbody8 or cluster" > gpurun_out/exp/t14.log 2>&1; tail -12 gpurun_out/exp/t14.log | cut -c1-300
