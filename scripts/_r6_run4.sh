mkdir -p gpurun_out/r6d
for it in 0 1 2 3; do GEN=scene POSE_ITERS=$it python scripts/f16_sweep_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6d/f16_sweep_bench_scene.txt; done
