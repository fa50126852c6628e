mkdir -p gpurun_out/exp
python profiles/experiments/dbg1.py 2>&1 | grep -E "n=128|n=4 |after|step|Error|error" | head -20
python profiles/experiments/exp_variant.py --tag v3 > gpurun_out/exp/v3.json 2> gpurun_out/exp/v3.err; cut -c1-700 gpurun_out/exp/v3.json; tail -n 3 gpurun_out/exp/v3.err
HEYOKA_AMD_PAIR_SPLIT=0 python profiles/experiments/exp_variant.py --tag v2 > gpurun_out/exp/v2.json 2> gpurun_out/exp/v2.err; cut -c1-700 gpurun_out/exp/v2.json
