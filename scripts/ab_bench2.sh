# like ab_bench.sh, printing the per-kernel totals of the timed call that the bench line carries
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/ab
for rep in 1 2; do for L in "$@"; do
  DICP_HIP_LIB=$R/$L timeout -k 10 240 python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/ab/out.json 2> $R/gpurun_out/ab/err.txt || { tail -3 $R/gpurun_out/ab/err.txt; exit 1; }
  python3 -c "
import json,sys; d=json.load(open('$R/gpurun_out/ab/out.json')); print('%-28s step %.4f ms  knn %.4f  bwd %.4f' % ('$L', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_streaming']['avg_launch_ms']))"
done; done
