# A/B of library builds at two shapes: bash scripts/ab_libs2.sh <lib.so> ...   ("-" = the in-tree build)
for rnd in 1 2; do for L in "$@"; do
  if [ "$L" = "-" ]; then unset DICP_HIP_LIB; else export DICP_HIP_LIB=$PWD/$L; fi
  python bench.py --no-cpu-baseline --no-extra-legs | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'B=256x16384', round(d['value']), 'knn', round(d['roofline']['avg_launch_ms'],4), 'acc', round(d['roofline_accumulate']['avg_launch_ms'],4), 'bwd', round(d['roofline_streaming']['avg_launch_ms'],4))"
  python bench.py --no-cpu-baseline --no-extra-legs --batch 64 --points 65536 --steps 5 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'B=64x65536 K=5', round(d['value']), 'knn', round(d['roofline']['avg_launch_ms'],4), 'acc', round(d['roofline_accumulate']['avg_launch_ms'],4), 'bwd', round(d['roofline_streaming']['avg_launch_ms'],4))"
done; done
