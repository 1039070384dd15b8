#!/bin/bash
# dev tool: VGPR / spill counts of the photometric linearize instantiations in an object file (scripts/build_variant.sh leaves /tmp/sage_var/<name>.o)
#   scripts/kregs.sh /tmp/sage_var/libsage_x.o [kernel-substring]
O=$1; PAT=${2:-photo_kernelILi32ELi16ELb1ELi2}
D=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$D/fat.bin $O $D/o.tmp 2>/dev/null
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$D/fat.bin --output=$D/dev.o --unbundle
$B/llvm-readelf --notes $D/dev.o | awk -v pat="$PAT" '
/\.name:/ {name=$2}
/\.sgpr_spill_count:/ {ss=$2}
/\.vgpr_count:/ {vg=$2}
/\.vgpr_spill_count:/ {vs=$2; if (index(name, pat)) print name, "vgpr", vg, "vgpr_spill", vs, "sgpr_spill", ss}
'
rm -rf $D
