#!/bin/bash
# Counts quarter-rate integer multiplies (32-bit lo/hi, 64-bit mad) per kernel in the device ISA.
S=${1:-build/circl_hip-hip-amdgcn-amd-amdhsa-gfx950.s}
awk '/^_Z[A-Za-z0-9_]*:/ {name=$1; sub(":","",name)}
     /^[ \t]+v_(mul_lo_u32|mul_hi_u32|mul_hi_i32|mad_u64_u32|mad_i64_i32) / {slow[name]++}
     /^[ \t]+v_/ {valu[name]++}
     END {for (k in valu) printf "%6d %6d %s\n", slow[k], valu[k], k}' "$S" | sort -rn | c++filt | cut -c1-150
