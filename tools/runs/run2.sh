set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_primitives_gpu.py -x -q -k "gemm_row" > gpurun_out/r04/gputest_row.log 2>&1 || { tail -40 gpurun_out/r04/gputest_row.log; exit 1; }
tail -3 gpurun_out/r04/gputest_row.log
timeout -k 10 300 python tools/gemm_row_bench.py > gpurun_out/r04/gemm_row_bench_0.txt 2>&1 || { tail -20 gpurun_out/r04/gemm_row_bench_0.txt; exit 1; }
cat gpurun_out/r04/gemm_row_bench_0.txt
python -m pytest tests -m gpu -q > gpurun_out/r04/gputest_2.log 2>&1 || { tail -60 gpurun_out/r04/gputest_2.log; }
tail -15 gpurun_out/r04/gputest_2.log
python tools/gemm_shapes.py comer > gpurun_out/r04/gemm_shapes_comer_0.txt 2>&1
python tools/gemm_shapes.py > gpurun_out/r04/gemm_shapes_base_0.txt 2>&1
python tools/comer_bench.py > gpurun_out/r04/comer_bench_0.txt 2>&1
tail -2 gpurun_out/r04/comer_bench_0.txt
