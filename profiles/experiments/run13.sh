mkdir -p gpurun_out/exp
HEYOKA_AMD_WAVE_ROLES=1 python profiles/experiments/exp_variant.py --tag v4 --dt 40 --calls 3 > gpurun_out/exp/v4.json 2> gpurun_out/exp/v4.err; cut -c1-600 gpurun_out/exp/v4.json; tail -n 2 gpurun_out/exp/v4.err
python profiles/experiments/exp_variant.py --tag v3 --dt 40 --calls 3 > gpurun_out/exp/v3l.json 2> gpurun_out/exp/v3l.err; cut -c1-600 gpurun_out/exp/v3l.json
