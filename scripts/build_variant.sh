# A/B builds of the library with extra -D flags (for DICP_HIP_LIB=...): dicp_kernels.hip is recompiled, the other two objects are the tree's.
# usage: bash scripts/build_variant.sh <name> [-DX=Y ...]   ->  dicp_amd/_variants/libdicp_<name>.so   (git-ignored, travels with gpurun)
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
mkdir -p dicp_amd/_variants
FL=$(python3 -c "from dicp_amd import _lib; print(' '.join(_lib.FLAGS))")
/opt/rocm/bin/hipcc $FL "$@" -I include -c -o dicp_amd/_variants/k_$name.o dicp_amd/csrc/dicp_kernels.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dicp_amd/_variants/libdicp_$name.so dicp_amd/_variants/k_$name.o dicp_amd/csrc/_obj/knn_f16.o dicp_amd/csrc/_obj/dicp_call.o || exit 1
rm -f dicp_amd/_variants/k_$name.o
echo built dicp_amd/_variants/libdicp_$name.so
