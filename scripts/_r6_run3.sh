set -e
mkdir -p gpurun_out/r6c
for it in 0 1 2 3; do POSE_ITERS=$it python scripts/f16_sweep_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6c/f16_sweep_bench.txt; done
python -m pytest tests/test_gpu_parity.py -k "query_order" -x -q 2>&1 | tail -5
