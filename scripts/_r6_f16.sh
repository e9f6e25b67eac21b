mkdir -p gpurun_out/r6g
for gen in pairs indep; do for it in 0 1; do GEN=$gen POSE_ITERS=$it python scripts/f16_sweep_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6g/f16_sweep_bench.txt; done; done
B=64 NPTS=65536 POSE_ITERS=0 python scripts/f16_sweep_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6g/f16_sweep_bench.txt
python -m pytest tests/test_gpu_f16.py -x -q 2>&1 | tail -2
