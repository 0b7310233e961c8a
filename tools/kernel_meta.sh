#!/bin/bash
# tools/kernel_meta.sh [pattern] -- registers, scratch and LDS of the kernels in libg1s_diff.so (ISA metadata of the gfx950 code object)
P=${1:-k3s_fused}
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $(dirname $0)/../grav1synth_amd/libg1s_diff.so $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/co.o --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/co.o | grep -E "\.name:|\.vgpr_count|\.sgpr_count|private_segment_fixed_size|group_segment_fixed|vgpr_spill|sgpr_spill" | paste - - - - - - - | grep -E "$P" | sed 's/ \+/ /g; s/\t/ /g' | c++filt | sed 's/(g1s::Geom, g1s::FParams)//'
[ -n "$2" ] && cp $T/co.o $2
rm -rf $T
