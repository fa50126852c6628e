set -x
mkdir -p gpurun_out/exp
P=profiles/experiments/exp_variant.py
python $P --tag base > gpurun_out/exp/base.json 2> gpurun_out/exp/base.err
HEYOKA_AMD_V2_BS=512 HEYOKA_AMD_JET_GLOBAL=1 python $P --tag bs512_jetglobal > gpurun_out/exp/bs512.json 2> gpurun_out/exp/bs512.err
python $P --tag base_o11 --tol 1e-8 > gpurun_out/exp/base_o11.json 2> gpurun_out/exp/base_o11.err
HEYOKA_AMD_V2_BS=512 python $P --tag bs512_o11 --tol 1e-8 > gpurun_out/exp/bs512_o11.json 2> gpurun_out/exp/bs512_o11.err
python $P --tag base_o7 --tol 1e-5 > gpurun_out/exp/base_o7.json 2> gpurun_out/exp/base_o7.err
HEYOKA_AMD_V2_BS=512 python $P --tag bs512_o7 --tol 1e-5 > gpurun_out/exp/bs512_o7.json 2> gpurun_out/exp/bs512_o7.err
cat gpurun_out/exp/*.json | cut -c1-600
tail -3 gpurun_out/exp/*.err
