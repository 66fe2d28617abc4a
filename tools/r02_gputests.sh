set -e
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/gputests.log 2>&1 || { tail -40 gpurun_out/r02/gputests.log; exit 1; }
tail -3 gpurun_out/r02/gputests.log
