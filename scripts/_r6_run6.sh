mkdir -p gpurun_out/r6f
python -m pytest tests/test_gpu_hints.py tests/test_gpu_config2.py tests/test_gpu_parity.py -x -q -k "hint or config2 or gauss or svd_step or query or frame or rotating" > gpurun_out/r6f/pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r6f/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash scripts/call_timeline.sh r6f/tl_k10 random 10 8 > /dev/null 2>&1
cut -c1-120 gpurun_out/r6f/tl_k10/timeline.txt | awk '{printf "%s %s %s\n", $1, $4, $8 $9 $10}' | head -32
tail -1 gpurun_out/r6f/tl_k10/timeline.txt
