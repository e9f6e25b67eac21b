mkdir -p gpurun_out/r04a
for f in 0 1 2 3; do echo "== DICP_F16_FORM=$f"; DICP_F16_FORM=$f timeout -k 10 200 python scripts/knn_f16_check.py 2>&1 | grep -v amdgpu.ids | head -9; done > gpurun_out/r04a/knn_f16_forms.txt 2>&1
cat gpurun_out/r04a/knn_f16_forms.txt
