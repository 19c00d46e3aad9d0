#!/usr/bin/env python3
"""kernel_resources.py <lib.so or .o> [name filter] -> VGPRs, SGPRs, scratch, LDS, spills of every gfx950 kernel in the fat binary
(the code objects' .note metadata; what hipcc -Rpass-analysis=kernel-resource-usage prints, after the fact)."""
import os, re, struct, subprocess, sys, tempfile
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
W = tempfile.mkdtemp()
fat = os.path.join(W, "fat.bin")
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
b = open(fat, "rb").read()
pos, k = 0, 0
while True:
    i = b.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
    if i < 0:
        break
    n = struct.unpack_from("<Q", b, i + 24)[0]
    p = i + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", b, p); p += 24
        triple = b[p:p + tl].decode(); p += tl
        if "gfx950" in triple and size:
            co = os.path.join(W, "co_%d.o" % k); k += 1
            open(co, "wb").write(b[i + off:i + off + size])
            out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in out.split("- .agpr_count")[1:]:
                g = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, "?"])[1]
                name = g("name")
                if flt in name:
                    print("%-70s vgpr %3s sgpr %3s scratch %5s lds %6s vgpr_spill %3s sgpr_spill %3s" % (name[:70], g("vgpr_count"), g("sgpr_count"),
                          g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
    pos = i + 24
