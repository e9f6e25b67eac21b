export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/gputest.log
timeout -k 10 300 python scripts/soak_modes.py 1000 > gpurun_out/soak_modes.txt 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/soak_modes.txt
