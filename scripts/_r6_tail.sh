mkdir -p gpurun_out/r6tail
for mode in 0 1; do
  for tol in "" 1e-4; do
    DICP_TAIL_ALL=$mode DICP_TOL=$tol bash scripts/call_timeline.sh r6tail/tl_${mode}_${tol:-k10} random 10 8 > /dev/null 2>&1
    echo "TAIL_ALL=$mode tol=${tol:-none}"; grep -E "bwd_tail|accumulate_bwd_window|step_bwd|window_reduce|call:" gpurun_out/r6tail/tl_${mode}_${tol:-k10}/timeline.txt | awk '{n[$NF]++; d[$NF]+=$4} END {for (k in n) printf "   %-50s %3d %9.1f\n", k, n[k], d[k]}'
    tail -1 gpurun_out/r6tail/tl_${mode}_${tol:-k10}/timeline.txt
  done
done
