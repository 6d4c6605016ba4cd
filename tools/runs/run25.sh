set -e
mkdir -p gpurun_out/r04
python tools/find_small_launches.py > gpurun_out/r04/small_base.txt 2>&1 || true
python tools/find_small_launches.py comer > gpurun_out/r04/small_comer.txt 2>&1 || true
grep " x " gpurun_out/r04/small_base.txt | wc -l; grep " x " gpurun_out/r04/small_comer.txt | wc -l
