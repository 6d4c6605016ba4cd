set -e
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest_1.log 2>&1 || { tail -40 gpurun_out/r04/gputest_1.log; exit 1; }
tail -5 gpurun_out/r04/gputest_1.log
python tools/gemm_shapes.py comer > gpurun_out/r04/gemm_shapes_comer_0.txt 2>&1
python tools/gemm_shapes.py > gpurun_out/r04/gemm_shapes_base_0.txt 2>&1
python tools/comer_bench.py > gpurun_out/r04/comer_bench_0.txt 2>&1
tail -2 gpurun_out/r04/comer_bench_0.txt
