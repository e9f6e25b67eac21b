mkdir -p gpurun_out/r6full
python -m pytest tests -m gpu -x -q > gpurun_out/r6full/pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r6full/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6full/bench_k20.json 2> gpurun_out/r6full/bench_k20.err; rc=$?
tail -c 600 gpurun_out/r6full/bench_k20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6full/bench_k20.json").read().strip().splitlines()[-1])
keys = ["value", "value_k10", "value_tolerance", "value_structured", "value_independent", "value_bruteforce", "ms_per_step"]
print({k: d.get(k) for k in keys})
print("indep", json.dumps(d.get("independent"))[:1500])
print("roofline", d.get("roofline")); print("acc", d.get("roofline_accumulate")); print("stream", d.get("roofline_streaming"))
print("c4", d.get("value_c4"), d.get("value_c4_full"))
PY
exit $rc
