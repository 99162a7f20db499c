#!/bin/bash
# Resource figures of the gfx950 code object(s) inside a built unit: VGPRs, spills, scratch, LDS, and the counts of
# flat_ / scratch_ / global_ / ds_ instructions per kernel.   usage: tools/diag/codeobj.sh cave_amd/csrc/build/k_step.o
L=/opt/rocm/lib/llvm/bin
for obj in "$@"; do
  t=$(mktemp -d)
  $L/llvm-objcopy --dump-section .hip_fatbin=$t/fat.bin "$obj" 2>/dev/null || { echo "$obj: no .hip_fatbin"; continue; }
  tgt=$($L/clang-offload-bundler --list --type=o --input=$t/fat.bin | grep gfx950 | head -1)
  $L/clang-offload-bundler --unbundle --type=o --input=$t/fat.bin --targets="$tgt" --output=$t/dev.o
  echo "== $obj ($tgt)"
  $L/llvm-readelf --notes $t/dev.o | grep -E "^ +\.name:|\.vgpr_count|\.sgpr_count|spill_count|private_segment_fixed|\.group_segment_fixed" | sed 's/^ *//' | paste -sd' ' | sed 's/\.name:/\n.name:/g'
  $L/llvm-objdump -d $t/dev.o > $t/dis.txt
  awk '/^[0-9a-f]+ <.*>:$/ {name=$2} /\t(flat_|scratch_|global_|ds_|v_readlane|v_writelane|s_nop)/ {split($0,a,"\t"); split(a[2],b," "); k=b[1]; sub(/_.*/,"_",k); if (b[1] ~ /^v_readlane/) k="v_readlane"; if (b[1] ~ /^v_writelane/) k="v_writelane"; if (b[1] ~ /^s_nop/) k="s_nop"; cnt[name" "k]++} END {for (x in cnt) print cnt[x], x}' $t/dis.txt | sort -k2,2 -k3,3 | awk '{printf "  %-12s %6d  %s\n", $3, $1, $2}' | cut -c1-150
  ls -la $t/dev.o | awk '{print "  code object bytes:", $5}'
  rm -rf $t
done
