set -e
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest_32.log 2>&1 || { tail -40 gpurun_out/r04/gputest_32.log; exit 1; }
tail -2 gpurun_out/r04/gputest_32.log
