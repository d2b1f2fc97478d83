#!/bin/bash
# Register / scratch / LDS use of every kernel of the library, from the code objects embedded in csrc/build/*.o (build host only).
B=${1:-/root/repo/eonerf_code_amd/csrc/build}
LL=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
for o in $B/*.o; do
  $LL/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $o 2>/dev/null || continue
  $LL/clang-offload-bundler --unbundle --input=$TMP/fat.bin --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/dev.co 2>/dev/null || continue
  $LL/llvm-readelf --notes $TMP/dev.co | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = re.sub(r"_ZN\d+_GLOBAL__N_1", "", g("name"))
    print("%-100s vgpr %4s agpr %4s sgpr %4s scratch %5s lds %6s" % (name[:100], g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
'
done
rm -rf $TMP
