mkdir -p gpurun_out/exp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_models.py -x -q -m gpu -k "outer_ss or cluster or nbody8 or models or random_systems or write_tc or tutorial or device_array" > gpurun_out/exp/t2.log 2>&1; tail -5 gpurun_out/exp/t2.log
python profiles/experiments/exp_variant.py --tag b1 > gpurun_out/exp/b1.json 2> gpurun_out/exp/b1.err; cut -c1-700 gpurun_out/exp/b1.json; tail -n 3 gpurun_out/exp/b1.err
