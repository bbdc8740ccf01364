#!/bin/bash
# Disassembly of one kernel of a built engine library: tools/isa_of.sh lib.so <mangled-name-substring> > kernel.s
# (llvm-objdump --offloading extracts the device code object next to a copy of the library in /tmp/study)
mkdir -p /tmp/study
B=$(basename "$1")
cp "$1" /tmp/study/$B && (cd /tmp/study && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading $B > /dev/null 2>&1)
CO=$(ls /tmp/study/$B.*gfx950 2>/dev/null | head -1)
[ -n "$CO" ] || { echo "no gfx950 code object in $1" >&2; exit 1; }
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$CO" | awk -v k="$2" '/^[0-9a-f]+ <.*>:$/{f = index($0, k) > 0} f'
