mkdir -p gpurun_out/r6e
for gen in scene pairs; do for it in 0 1 2; do GEN=$gen POSE_ITERS=$it python scripts/f16_sweep_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6e/f16_sweep_bench.txt; done; done
GEN=indep POSE_ITERS=3 python scripts/f16_sweep_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6e/f16_sweep_bench.txt
python -m pytest tests/test_gpu_f16.py -x -q 2>&1 | tail -5
