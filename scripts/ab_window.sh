# A/B of window-kernel builds on ONE box: bash scripts/ab_window.sh <name> ...   (dicp_amd/_variants/libdicp_<name>.so, scripts/build_variant.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abw; mkdir -p $O
for L in "$@"; do
  export DICP_HIP_LIB=$R/dicp_amd/_variants/libdicp_$L.so
  timeout -k 10 240 python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-10} > $O/$L.json 2> $O/$L.err || { tail -n 5 $O/$L.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/$L.json').read().strip().splitlines()[-1]); s=d.get('roofline_streaming', d['roofline']); a=d.get('roofline_accumulate', d['roofline'])
print('%-14s value %8.0f  call_ms %s  bwd_window %.4f ms (frac %.3f)  acc %.4f ms' % ('$L', d['value'], d['call_ms'], s['avg_launch_ms'], s['frac'], a['avg_launch_ms']))"
  timeout -k 10 200 python3 $R/scripts/bwd_bench.py > $O/$L.bwd.txt 2>&1 || { tail -n 5 $O/$L.bwd.txt; exit 1; }
  grep -E "^window|far rows" $O/$L.bwd.txt | cut -c1-200
done
