mkdir -p gpurun_out/exp
timeout 600 python -m pytest tests/test_distributed.py tests/test_cpp_api.py -q -m gpu --timeout=300 > gpurun_out/exp/t10.log 2>&1; tail -4 gpurun_out/exp/t10.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/exp/bench10.json 2> gpurun_out/exp/bench10.err; tail -c 3000 gpurun_out/exp/bench10.json; tail -3 gpurun_out/exp/bench10.err
