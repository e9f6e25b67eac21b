mkdir -p gpurun_out/r6bench
python -m pytest tests/test_gpu_independent.py tests/test_gpu_f16.py tests/test_gpu_cert_soak.py -x -q > gpurun_out/r6bench/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r6bench/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6bench/bench_k20.json 2> gpurun_out/r6bench/bench_k20.err; rc=$?
tail -c 600 gpurun_out/r6bench/bench_k20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6bench/bench_k20.json").read().strip().splitlines()[-1])
keys = ["value", "value_k10", "value_tolerance", "value_structured", "value_independent", "value_bruteforce", "ms_per_step"]
print({k: d.get(k) for k in keys})
print("indep", {k: (v["cloud_it_per_s"], v.get("pairs_scored_fraction")) for k, v in d.get("independent", {}).items()})
print("roofline", {k: d["roofline"].get(k) for k in ("frac", "avg_launch_ms", "launch_ms_by_iteration")})
print("c4", d.get("value_c4", {}).get("sweep"), d.get("value_c4_full", {}).get("ms_per_iteration"))
print("c2", d.get("value_c2"))
PY
exit $rc
