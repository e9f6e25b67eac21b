mkdir -p gpurun_out/r6h
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_skip.py tests/test_gpu_independent.py tests/test_gpu_deterministic.py tests/test_gpu_ragged.py -x -q > gpurun_out/r6h/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r6h/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash scripts/call_timeline.sh r6h/indep_tl indep 10 6 > /dev/null 2>&1
grep -E "accumulate_bwd_window|call:" gpurun_out/r6h/indep_tl/timeline.txt | awk '{print $1, $4, $NF}'
DICP_TOL=1e-4 bash scripts/call_timeline.sh r6h/tol_tl random 10 6 > /dev/null 2>&1
grep -E "accumulate_bwd_window|call:" gpurun_out/r6h/tol_tl/timeline.txt | awk '{print $1, $4, $NF}'
