R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abad; mkdir -p $O
for rep in 1 2; do for A in 1 0; do
  DICP_F16_ADAPTIVE=$A timeout -k 10 240 python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-10} > $O/$A.json 2> $O/$A.err || { tail -n 5 $O/$A.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/$A.json').read().strip().splitlines()[-1])
print('adaptive $A  value %8.0f  call_ms %s  knn %s' % (d['value'], d['call_ms'], d['roofline']['launch_ms_by_iteration'][:4]))"
done; done
