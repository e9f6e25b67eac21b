R=$GRAFT_REPO_ROOT
for rep in 1 2; do for D in _ab_old .; do
  cd $R/$D; timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 > /tmp/o.json 2>/tmp/o.err || { tail -n 3 /tmp/o.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print('%-8s value %8.0f  call_ms %s  knn %s acc %s bwd %s' % ('$D', d['value'], d['call_ms'][:3], d['roofline']['launch_ms_by_iteration'][:4], d['roofline_accumulate']['launch_ms_by_iteration'][4:6], d['roofline_streaming']['launch_ms_by_iteration'][-3:]))"
done; done
