#!/bin/bash
# dev: registers / scratch of the kernels of one object file of the build (names filtered by a fragment)
# usage: tools/dev/kernel_regs.sh tsf_inst_quad.o [name-fragment]
O=$(dirname "$0")/../../time_series_spark_amd/_obj/$1
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section .hip_fatbin=/tmp/kr_fb.bin "$O" /tmp/kr_ignore.o || exit 1
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/kr_fb.bin --output=/tmp/kr.co --unbundle || exit 1
$L/llvm-readelf --notes /tmp/kr.co | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|private_segment_fixed_size|vgpr_spill_count|group_segment_fixed_size" | paste - - - - - - | sed 's/  */ /g' | grep "${2:-.}" | cut -c1-300
