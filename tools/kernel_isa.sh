#!/bin/bash
# ISA of one kernel of a compiled translation unit (CPU side): tools/kernel_isa.sh mixermdm_amd/csrc/gemm_f32.o '<mangled-name regex>' > out.s
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.o
$L/llvm-objdump -d --no-show-raw-insn $T/dev.o | awk -v pat="$2" '/^[0-9a-f]+ <.*>:$/ {on = ($0 ~ pat)} on {print}'
rm -rf $T
