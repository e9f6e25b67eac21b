# SQ counters of the sweep kernel (separate --pmc passes, kernel-trace only).  usage: bash scripts/pmc_sq.sh <outdir-name>
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_sq}; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "knn_sweep_kernel" in k or "knn_f16_sweep_kernel" in k or "accumulate_kernel" in k or "accumulate_bwd_window" in k:
            short = k.split("::")[1].split("(")[0] if "::" in k else k.split("(")[0][-60:]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/summary.txt", "w") as out:
    for k in sorted(agg):
        out.write(k + "\n")
        for c in sorted(agg[k]):
            v = agg[k][c]
            out.write("   %-28s n=%3d mean=%14.1f min=%14.1f max=%14.1f\n" % (c, len(v), sum(v) / len(v), min(v), max(v)))
print(open(sys.argv[1] + "/summary.txt").read())
PY
