#!/bin/bash
# usage: scratch/isa.sh <kernel-name-substring>  -> dumps ISA of that kernel to /tmp/somtemps/k.s and prints stats
set -e
mkdir -p /tmp/somtemps/b && cd /tmp/somtemps/b
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared /root/repo/xpysom_dask_amd/csrc/somhip.hip -o x.so -save-temps 2>/dev/null
S=$(ls *gfx950*.s | head -1)
L=$(grep -n "^_ZN6somhip.*$1.*:" $S | head -1 | cut -d: -f1)
E=$(awk -v s=$L 'NR>s && /s_endpgm/ {print NR; exit}' $S)
sed -n "${L},${E}p" $S > /tmp/somtemps/k.s
for p in v_mfma v_min3_i32 v_min_i32 v_and_or_b32 v_pk_add v_add_f32 v_mov_b32 ds_read_b128 v_accvgpr s_nop s_waitcnt v_cndmask v_cmp; do echo "$p $(grep -c $p /tmp/somtemps/k.s)"; done
grep -A12 "^_ZN6somhip.*$1.*\.num_vgpr\|\.set _ZN6somhip.*$1.*num_vgpr" $S | head -3
