#!/bin/bash
# tools/asm.sh <file.hip> <mangled-name-regex> : gfx950 assembly of one kernel to /tmp/asm/k.s
mkdir -p /tmp/asm
f=/root/repo/curvecloudnet_amd/csrc/$1
flags="-O3 --offload-arch=gfx950 -fPIC -std=c++17"
case "$1" in ccn_curve.hip|ccn_frnn.hip|ccn_sample.hip) flags="$flags -ffp-contract=off";; esac
(hipcc $flags -S --cuda-device-only -o /tmp/asm/full.s $f 2>/dev/null)
awk -v pat="^_Z.*$2.*:" '$0 ~ pat {p=1} p{print} /s_endpgm/{if(p){exit}}' /tmp/asm/full.s > /tmp/asm/k.s
wc -l /tmp/asm/k.s | awk '{print $1" lines"}'
grep -E "^\s+\.set .*$2.*\.(num_vgpr|numbered_sgpr)" /tmp/asm/full.s | head -2 | awk '{print $NF, $2}' | sed 's/_Z.*\.//'
