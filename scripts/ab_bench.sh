# A/B of two builds on ONE box: bash scripts/ab_bench.sh <libA.so> <libB.so>  (paths relative to the repo root)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/ab
for rep in 1; do for L in "$@"; do
  DICP_HIP_LIB=$R/$L timeout -k 10 240 python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/ab/out.json 2> $R/gpurun_out/ab/err.txt || { tail -3 $R/gpurun_out/ab/err.txt; exit 1; }
  python3 -c "
import json,sys; d=json.load(open('$R/gpurun_out/ab/out.json')); print('%-28s step %.4f ms  knn' % ('$L', d['ms_per_step']), d['roofline']['launch_ms_by_iteration'])"
done; done
