mkdir -p gpurun_out/exp
python profiles/experiments/exp_variant.py --tag v3c > gpurun_out/exp/v3c.json 2> gpurun_out/exp/v3c.err; cut -c1-600 gpurun_out/exp/v3c.json; tail -n 3 gpurun_out/exp/v3c.err
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_models.py -x -q -m gpu -k "outer_ss or cluster or nbody8 or models or random_systems or write_tc or tutorial or device_array or full_size" > gpurun_out/exp/t7.log 2>&1; tail -5 gpurun_out/exp/t7.log
