# A/B of environment switches on ONE box: bash scripts/env_ab.sh "VAR=a" "VAR=b" ...   ("-" = no override)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/ab
for rep in 1 2; do for E in "$@"; do
  if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
  env $EE timeout -k 10 240 python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/ab/out.json 2> $R/gpurun_out/ab/err.txt || { tail -3 $R/gpurun_out/ab/err.txt; exit 1; }
  python3 -c "
import json,sys; d=json.load(open('$R/gpurun_out/ab/out.json')); print('%-28s step %.4f ms  knn' % ('$E', d['ms_per_step']), d['roofline']['launch_ms_by_iteration'])"
done; done
