#!/bin/bash
# Registers / scratch / LDS of every kernel in the built objects (build/core/*.o): the occupancy inputs quoted in DESIGN.md.
# usage: profiles/kernel_resources.sh [object ...]
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
for o in "${@:-build/core/*.o}"; do
  for f in $o; do
    $B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$f" 2>/dev/null || continue
    $B/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
    $B/llvm-readelf --notes $T/k.co | awk '
      /\.lds_size:/ {lds=$2} /\.name:/ {name=$0; sub(/.*\.name: */,"",name)} /\.private_segment_fixed_size:/ {scr=$2}
      /\.sgpr_count:/ {sg=$2} /\.agpr_count:/ {ag=$2}
      /\.vgpr_count:/ {printf "%-70s vgpr %3d agpr %3d sgpr %3d scratch %4d B\n", name, $2, ag, sg, scr}' | c++filt | sed 's/tg:://g; s/(tg::DScene.*//'
  done
done
rm -rf $T
