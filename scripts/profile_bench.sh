cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_final; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/write.log 2>&1
python3 $R/scripts/pmc_summary.py $O/fetch $O/write $O/pmc_hbm_traffic.json 256 16384 > /dev/null
ls $O $O/stats | head -20
