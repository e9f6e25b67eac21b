set -x
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/gputest.log
