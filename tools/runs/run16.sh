set -e
R=$PWD
mkdir -p gpurun_out/r04
bash tools/refresh_profiles.sh r04
ls -la gpurun_out/r04* | head
python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
tail -c 1500 gpurun_out/r04/bench_default.json
