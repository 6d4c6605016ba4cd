set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests/test_primitives_gpu.py tests/test_comer_gpu.py -x -q > gpurun_out/r04/gputest_18.log 2>&1 || { tail -40 gpurun_out/r04/gputest_18.log; exit 1; }
tail -2 gpurun_out/r04/gputest_18.log
python tools/gemm_row_bench.py > gpurun_out/r04/gemm_row_bench_4.txt 2>&1; cat gpurun_out/r04/gemm_row_bench_4.txt
python tools/comer_bench.py > gpurun_out/r04/comer_bench_13.txt 2>&1; tail -1 gpurun_out/r04/comer_bench_13.txt
