# Register / scratch / LDS footprint of every kernel of libdicp_hip.so, from the compiler's own remarks (runs without a GPU).
# usage: bash scripts/kernel_resources.sh [name-filter-regex] > profiles/rNN_kernel_resources.txt
cd "$(dirname "$0")/.." || exit 1
FLT=${1:-.}
for f in dicp_kernels knn_f16; do
  /opt/rocm/bin/hipcc $(python3 -c "from dicp_amd import _lib; print(' '.join(_lib.FLAGS))") -I include -c -o /dev/null dicp_amd/csrc/$f.hip \
      -Rpass-analysis=kernel-resource-usage 2>&1 | python3 scripts/kernel_resources.py "$FLT"
done
