# rocprofv3 passes of the benchmark command (timed workload only: no extra legs, no CPU baseline), each in its own run as the guide
# prescribes (kernel-trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE).  usage: bash scripts/profile_bench.sh <out-name> <commit>
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof}; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs > $O/write.log 2>&1
python3 $R/scripts/pmc_summary.py $O/fetch $O/write $O/pmc_hbm_traffic.json 256 16384 ${2:-unknown} > /dev/null
python3 $R/scripts/kstats.py $O/stats 30 > $O/kernel_stats.txt
python3 $R/scripts/trace_timeline.py $O/stats/s_kernel_trace.csv --full > $O/timeline.txt
cp $O/stats/s_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/stats $O/fetch $O/write
ls $O
