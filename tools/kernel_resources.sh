#!/bin/bash
# Registers, spills and LDS of every kernel in an object file of the build (no GPU needed): tools/kernel_resources.sh build/obj/dp_kernels.hip.o [filter]
set -e
obj=${1:-build/obj/dp_kernels.hip.o}
filter=${2:-.}
tmp=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin "$obj"
$L/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/k.co
$L/llvm-readelf --notes $tmp/k.co | grep -E "^ +\.name:|\.vgpr_count|\.agpr_count|vgpr_spill|sgpr_spill|group_segment_fixed|\.sgpr_count" | sed 's/^ *//' | awk '/^\.agpr_count/{a=$2} /^\.group_segment/{l=$2} /^\.name:/{n=$2} /^\.sgpr_count/{s=$2} /^\.sgpr_spill/{ss=$2} /^\.vgpr_count/{v=$2} /^\.vgpr_spill/{print n, "vgpr", v, "agpr", a, "sgpr", s, "lds", l, "vspill", $2, "sspill", ss}' | c++filt | grep -E "$filter" || true
rm -rf $tmp
