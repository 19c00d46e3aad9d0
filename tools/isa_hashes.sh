#!/bin/bash
# isa_hashes.sh <lib.so> -> one line per kernel: md5 of its disassembled gfx950 instructions (addresses stripped), kernel name.
# Used to check that a refactoring left the kernels' machine code unchanged (profiles/r04/isa_hashes_*.txt).
set -e
LIB=$(readlink -f "$1"); W=$(mktemp -d); cd $W
OBJDUMP=/opt/rocm/lib/llvm/bin/llvm-objdump
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$LIB >/dev/null 2>&1 || true
# the fat binary sits in section .hip_fatbin: pull every gfx950 code object out of it
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $LIB fat.bin
python3 - <<'PY'
import re
b = open('fat.bin', 'rb').read()
# concatenated "__CLANG_OFFLOAD_BUNDLE__" containers (one per translation unit)
k = 0
pos = 0
while True:
    i = b.find(b'__CLANG_OFFLOAD_BUNDLE__', pos)
    if i < 0: break
    import struct
    n = struct.unpack_from('<Q', b, i + 24)[0]
    p = i + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from('<QQQ', b, p); p += 24
        triple = b[p:p + tl].decode(); p += tl
        if 'gfx950' in triple and size:
            open('co_%d.o' % k, 'wb').write(b[i + off:i + off + size]); k += 1
    pos = i + 24
PY
for f in co_*.o; do
  $OBJDUMP -d --no-show-raw-insn $f | awk '/^[0-9a-f]+ <.*>:$/ {name=$2; next} name!="" && NF {sub(/^ +[0-9a-f]+: */, ""); sub(/ *\/\/ [0-9A-F]+:.*$/, ""); print name "\t" $0}' >> all.txt
done
cut -f1 all.txt | uniq | sort -u | while read n; do
  printf "%s %s\n" "$(awk -F'\t' -v n="$n" '$1==n {print $2}' all.txt | md5sum | cut -d' ' -f1)" "$n"
done | sort -k2
rm -rf $W
