# A/B of library builds on one box: bash scripts/ab_libs.sh <lib.so> [<lib.so> ...]   ("-" = the in-tree build)
for rnd in 1 2; do for L in "$@"; do
  if [ "$L" = "-" ]; then unset DICP_HIP_LIB; else export DICP_HIP_LIB=$PWD/$L; fi
  python bench.py --no-cpu-baseline --no-extra-legs | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), d['call_ms'], [round(v,3) for v in d['roofline']['launch_ms_by_iteration'][3:7]], [round(v,3) for v in d['roofline_accumulate']['launch_ms_by_iteration'][3:7]], d['roofline']['certified_iterations']['queries_searched_again'][4:6])"
done; done
