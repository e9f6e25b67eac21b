export TMPDIR=/tmp
timeout -k 10 500 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_k20.json 2> gpurun_out/bench_k20.err; echo "k20 rc=$?"
timeout -k 10 400 python bench.py > gpurun_out/bench_k10.json 2> gpurun_out/bench_k10.err; echo "k10 rc=$?"
python scripts/headline_host_time.py 20 30 | tail -1
