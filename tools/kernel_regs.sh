#!/bin/bash
# registers / spills / code size of the kernels in a hipcc object: tools/kernel_regs.sh serl_amd/csrc/build/rollout_team_nominal.o [name filter]
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | grep -E "\.name:|\.vgpr_count|\.vgpr_spill|\.sgpr_spill|private_segment_fixed|group_segment_fixed" | paste - - - - - - | grep "${2:-kernel}" | sed 's/  */ /g'
/opt/rocm/lib/llvm/bin/llvm-readelf -s $T/k.co | awk '$4=="FUNC"{print $3, $8}' | grep "${2:-kernel}" | sort -u
rm -rf $T
