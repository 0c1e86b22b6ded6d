#!/bin/bash
# kasm.sh <libdbgk.so> <kernel-name-regex> [out.s]: disassembly of one gfx950 kernel of the library + its resource figures
# (llvm-objdump --offloading unbundles the code objects next to a copy of the library in a scratch directory)
set -e
LLVM=/opt/rocm/lib/llvm/bin
W=$(mktemp -d)
cp "$1" $W/lib.so
(cd $W && $LLVM/llvm-objdump --offloading lib.so >/dev/null 2>&1)
for co in $W/lib.so.*gfx950; do
  $LLVM/llvm-objdump -d --no-show-raw-insn $co > $W/all.s
  if grep -qE "^[0-9a-f]+ <.*$2" $W/all.s; then
    awk -v pat="$2" '/^[0-9a-f]+ </{on = ($0 ~ pat)} on' $W/all.s | sed -e 's/\s*\/\/ [0-9A-F]*:.*$//' -e 's/<_Z[^>]*+\(0x[0-9a-f]*\)>/<+\1>/' > ${3:-/dev/stdout}.tmp
    mv ${3:-/dev/stdout}.tmp ${3:-$W/k.s}
    f=${3:-$W/k.s}
    echo "VALU $(grep -cE '^\s+v_' $f)  SALU $(grep -cE '^\s+s_' $f)  LDS $(grep -cE '^\s+ds_' $f)  VMEM $(grep -cE '^\s+(global|buffer|flat|scratch)_' $f)  lines $(wc -l < $f)"
    $LLVM/llvm-readelf --notes $co | awk -v pat="$2" '/\.name:/{on = ($0 ~ pat)} on && /vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|\.name:/' | head -12
  fi
done
rm -rf $W
