#!/bin/bash
# round 4: bit-identity of the touched families, then interleaved A/B against the round-3 library
set -e
mkdir -p gpurun_out/r04
L=flashattention-from-scratch-with-triton_amd/libmi355fa.so
timeout -k 10 300 python tools/check_family.py dq 3 4 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r04/check_dq_3_4.txt
timeout -k 10 300 python tools/check_family.py dkv 2 3 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r04/check_dkv_2_3.txt
timeout -k 10 200 python tools/kbench.py --libs ab/r3base.so@4,3,3,$L@4,4,3 --kernels fwd,dq,dkv --rounds 7 --reps 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/kb2_causal.txt
timeout -k 10 200 python tools/kbench.py --libs ab/r3base.so@4,3,3,$L@4,4,3 --kernels fwd,dq,dkv --rounds 7 --reps 10 --non-causal 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/kb2_full.txt
timeout -k 10 200 python tools/kbench.py --libs ab/r3base.so@4,1,2,$L@4,1,2 --kernels fwd --dim 128 --rounds 5 --reps 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/kb2_d128.txt
