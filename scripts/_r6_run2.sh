set -e
mkdir -p gpurun_out/r6b
python scripts/indep_sweep_forms.py 256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6b/indep_sweep_forms.txt
python scripts/indep_forms.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6b/indep_forms.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_f16.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r6b/pytest.log 2>&1 || (tail -40 gpurun_out/r6b/pytest.log; exit 1)
tail -3 gpurun_out/r6b/pytest.log
