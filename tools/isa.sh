#!/bin/bash
# Disassemble one kernel of a built object:  bash tools/isa.sh wfx_stages 'notch_kernelIs' > /tmp/k.s
# (extracts the gfx950 code object from the fat object first; prints the register / LDS usage on stderr)
set -e
B=wefax_amd/csrc/build
LL=/opt/rocm/lib/llvm/bin
rm -f $B/$1.o.*.hipv4-* $B/$1.o.*.host-*
(cd $B && $LL/llvm-objdump --offloading $1.o > /dev/null)
CO=$(ls $B/$1.o.*.hipv4-amdgcn-amd-amdhsa--gfx950 | head -1)
SYM=$($LL/llvm-readelf --notes "$CO" | grep -E "^\s+\.name:" | awk '{print $2}' | grep -E "$2" | head -1)
$LL/llvm-readelf --notes "$CO" | grep -E "^\s+\.(name|vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count):" | sed 's/ \+/ /g' | paste - - - - - - | grep -F " $SYM" >&2
$LL/llvm-objdump -d "$CO" --disassemble-symbols="$SYM" | cut -f2 | cut -d/ -f1 | sed 's/ \+$//'
rm -f $B/$1.o.*.hipv4-* $B/$1.o.*.host-*
