set -e
R=$PWD
P=weclip-vit-comer_amd
mkdir -p gpurun_out/r04
for rep in 1 2; do
for v in old new; do
cp $P/libweclip_hip_$v.so $P/libweclip_hip.so
echo "== $v ($rep)"
python tools/gemm_bench.py 2>&1 | grep "TF/s"
done
done > gpurun_out/r04/gemm_ab.txt
cat gpurun_out/r04/gemm_ab.txt
cp $P/libweclip_hip_new.so $P/libweclip_hip.so
python -m pytest tests/test_primitives_gpu.py tests/test_weclip_gpu.py tests/test_bench_size_golden_gpu.py -x -q > gpurun_out/r04/gputest_17.log 2>&1 || { tail -40 gpurun_out/r04/gputest_17.log; exit 1; }
tail -2 gpurun_out/r04/gputest_17.log
for v in old new old new; do
cp $P/libweclip_hip_$v.so $P/libweclip_hip.so
python bench.py --repeats 3 --no-cpu-baseline --no-extras > gpurun_out/r04/bench_ab_$v.json 2> gpurun_out/r04/bench_ab_$v.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04/bench_ab_$v.json').read().strip().splitlines()[-1])
print('$v', {k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')}, d['roofline']['frac'], d['roofline']['avg_launch_us'])
PY
done
