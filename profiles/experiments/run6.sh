mkdir -p gpurun_out/exp
python profiles/experiments/dbg1.py 2>&1 | grep -E "n=128|after|Error|error" | head -12
python profiles/experiments/exp_variant.py --tag v3b > gpurun_out/exp/v3b.json 2> gpurun_out/exp/v3b.err; cut -c1-600 gpurun_out/exp/v3b.json; tail -n 3 gpurun_out/exp/v3b.err
