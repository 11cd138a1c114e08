#!/bin/bash
# round 4: first GPU contact of the family-4 dQ kernel: bit-identity against family 3, then interleaved A/B
set -e
mkdir -p gpurun_out/r04
L=flashattention-from-scratch-with-triton_amd/libmi355fa.so
timeout -k 10 300 python tools/check_family.py dq 3 4 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/r04/check_dq_3_4.txt
timeout -k 10 200 python tools/kbench.py --libs $L@0,3,0,$L@0,4,0 --kernels dq --rounds 5 --reps 10 --non-causal 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/kb_dq4_full.txt
timeout -k 10 200 python tools/kbench.py --libs $L@0,3,0,$L@0,4,0 --kernels dq --rounds 5 --reps 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/kb_dq4_causal.txt
