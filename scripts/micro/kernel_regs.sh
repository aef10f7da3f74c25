#!/bin/bash
# registers / spills / LDS of every kernel in an object of the library:  scripts/micro/kernel_regs.sh ide-3d_amd/lib/obj/modconv.o [grep pattern]
O=$1; P=${2:-.}
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $O $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.o | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.agpr_count:/{a=$2} /\.vgpr_spill_count:/{s=$2} /\.group_segment_fixed_size:/{l=$2} /\.wavefront_size:/{print n, "vgpr", v, "agpr", a, "spill", s, "lds", l}' | c++filt | grep -E "$P"
rm -rf $T
