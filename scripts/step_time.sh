# mean duration of the step kernels in mid-size, small and headline calls (rocprofv3 --kernel-trace).  usage: bash scripts/step_time.sh <out-name>
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-step}; mkdir -p $O
rocprofv3 --kernel-trace -d $O/a -o t --output-format csv -- python3 $R/scripts/call_host_time.py > $O/a.log 2>&1
rocprofv3 --kernel-trace -d $O/b -o t --output-format csv -- python3 $R/scripts/small_call_profile.py > $O/b.log 2>&1
rocprofv3 --kernel-trace -d $O/c -o t --output-format csv -- python3 $R/scripts/call_profile.py random 20 6 > $O/c.log 2>&1
cd $R
for d in a b c; do echo "== $d"; python3 scripts/kstats.py $O/$d 40 | grep "step\|small\|kernel  "; done
grep median $O/b.log
