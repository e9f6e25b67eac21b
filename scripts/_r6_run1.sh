set -e
mkdir -p gpurun_out/r6a
python -m pytest tests/test_gpu_parity.py -k "search_frame" -x -q > gpurun_out/r6a/pytest_frame.log 2>&1 || (tail -30 gpurun_out/r6a/pytest_frame.log; exit 1)
python scripts/frame_directions.py 64 > gpurun_out/r6a/frame_directions.txt 2>&1 || (tail -30 gpurun_out/r6a/frame_directions.txt; exit 1)
cat gpurun_out/r6a/frame_directions.txt
python scripts/indep_forms.py > gpurun_out/r6a/indep_forms.txt 2>&1 || (tail -30 gpurun_out/r6a/indep_forms.txt; exit 1)
cat gpurun_out/r6a/indep_forms.txt
bash scripts/call_timeline.sh r6a/indep_tl indep 10 5
python -m pytest tests/test_gpu_independent.py tests/test_gpu_ragged.py -x -q > gpurun_out/r6a/pytest_indep.log 2>&1 || (tail -30 gpurun_out/r6a/pytest_indep.log; exit 1)
tail -3 gpurun_out/r6a/pytest_indep.log
