# Timing builds of the matrix-core sweep that stop after a stage (DICP_F16_ABLATE = 1 prologue, 2 + the sweep, 3 + margins and the winners' rows): where its time goes.
# usage (on the GPU box): bash scripts/f16_ablate.sh [GEN] [POSE_ITERS]      (builds need hipcc: run scripts/f16_ablate.sh build HERE first)
cd "$(dirname "$0")/.." || exit 1
if [ "$1" = "build" ]; then
  mkdir -p dicp_amd/_variants
  FL=$(python3 -c "from dicp_amd import _lib; print(' '.join(_lib.FLAGS))")
  for a in 1 2 3; do
    /opt/rocm/bin/hipcc $FL -DDICP_F16_ABLATE=$a -I include -c -o dicp_amd/_variants/f16_$a.o dicp_amd/csrc/knn_f16.hip || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dicp_amd/_variants/libdicp_f16ablate$a.so dicp_amd/csrc/_obj/dicp_kernels.o dicp_amd/_variants/f16_$a.o dicp_amd/csrc/_obj/dicp_call.o || exit 1
    rm -f dicp_amd/_variants/f16_$a.o
  done
  exit 0
fi
for a in 1 2 3; do
  echo "stage $a:"; DICP_HIP_LIB=$PWD/dicp_amd/_variants/libdicp_f16ablate$a.so FORMS=mfma GEN=${1:-pairs} POSE_ITERS=${2:-0} python3 scripts/f16_sweep_bench.py 2>&1 | grep "^mfma"
done
echo "whole kernel:"; FORMS=mfma GEN=${1:-pairs} POSE_ITERS=${2:-0} python3 scripts/f16_sweep_bench.py 2>&1 | grep "^mfma"
