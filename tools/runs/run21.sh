set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests/test_primitives_gpu.py tests/test_torch_ops_gpu.py tests/test_weclip_gpu.py tests/test_comer_gpu.py -x -q > gpurun_out/r04/gputest_21.log 2>&1 || { tail -40 gpurun_out/r04/gputest_21.log; exit 1; }
tail -2 gpurun_out/r04/gputest_21.log
python tools/wgrad_bench.py > gpurun_out/r04/wgrad_bench_2.txt 2>&1; grep "TF/s" gpurun_out/r04/wgrad_bench_2.txt
python tools/comer_bench.py > gpurun_out/r04/comer_bench_15.txt 2>&1; tail -1 gpurun_out/r04/comer_bench_15.txt
