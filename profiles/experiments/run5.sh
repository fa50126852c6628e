mkdir -p gpurun_out/exp
python profiles/experiments/dbg1.py 2>&1 | grep -E "n=128|after|Error|error" | head -12
python profiles/experiments/exp_variant.py --tag v3m > gpurun_out/exp/v3m.json 2> gpurun_out/exp/v3m.err; cut -c1-700 gpurun_out/exp/v3m.json; tail -n 3 gpurun_out/exp/v3m.err
HEYOKA_AMD_V3_MERGED=0 python profiles/experiments/exp_variant.py --tag v3u > gpurun_out/exp/v3u.json 2> gpurun_out/exp/v3u.err; cut -c1-700 gpurun_out/exp/v3u.json
