#!/bin/bash
# round 4: the persistent / ring-continuing causal forward family 4 (ab/fwdp.so) against the build before it (ab/base.so)
set -e
mkdir -p gpurun_out/r04
python3 tools/check_libs.py ab/base.so ab/fwdp.so --force 4,0,0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/fwdp_check.txt
python3 tools/check_fwd4.py 4 2>&1 | tail -3 | tee gpurun_out/r04/fwdp_check_fwd4.txt
python3 tools/stamps_fwd4.py ab/stamps.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/stamps_fwd4_c.txt
python3 tools/kbench.py --libs ab/base.so,ab/fwdp.so --kernels fwd --impl 4,0,0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/fwdp_kb.txt
python3 tools/kbench.py --libs ab/base.so,ab/fwdp.so --kernels fwd --impl 1,0,0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/fwdp_kb.txt
python3 tools/kbench.py --libs ab/base.so,ab/fwdp.so --kernels fwd --impl 4,0,0 --dim 128 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/fwdp_kb.txt
python3 tools/kbench.py --libs ab/base.so,ab/fwdp.so --kernels fwd --impl 4,0,0 --batch 1 --heads 16 --seq 16384 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/fwdp_kb.txt
python3 tools/kbench.py --libs ab/base.so,ab/fwdp.so --kernels fwd --impl 4,0,0 --batch 8 --heads 32 --seq 2048 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/fwdp_kb.txt
