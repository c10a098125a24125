#!/bin/bash
# tools/disasm.sh <object.o> <out.s>: gfx950 disassembly of the device code inside a hipcc object file
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1" $T/x
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$L/llvm-objdump -d $T/dev.co > "$2"
rm -rf $T
