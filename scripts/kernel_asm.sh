# The gfx950 assembly of ONE kernel of dicp_kernels.hip / knn_f16.hip (runs without a GPU): bash scripts/kernel_asm.sh <file-stem> <mangled-name-regex> > out.s
cd "$(dirname "$0")/.." || exit 1
S=/tmp/dicp_$1.s
[ -f $S ] && [ $S -nt dicp_amd/csrc/$1.hip ] && [ -z "$(find dicp_amd/csrc include -newer $S -name '*.h')" ] || \
  /opt/rocm/bin/hipcc $(python3 -c "from dicp_amd import _lib; print(' '.join(_lib.FLAGS))") -I include -S -o $S dicp_amd/csrc/$1.hip --cuda-device-only 2>/dev/null
awk -v pat="^_ZN.*$2.*: " '$0 ~ pat {p=1} p{print} /^\.Lfunc_end/{if(p){exit}}' $S
