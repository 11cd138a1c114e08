#!/bin/bash
# round 4: FastDiv + carried next item in the three persistent kernels (ab/fdiv.so) against the build before (ab/base.so)
set -e
mkdir -p gpurun_out/r04
python3 tools/check_libs.py ab/base.so ab/fdiv.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/fdiv_check.txt
python3 tools/check_libs.py ab/base.so ab/fdiv.so --force 4,4,4 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/fdiv_check.txt
python3 tools/kbench.py --libs ab/base.so,ab/fdiv.so --kernels fwd,dq,dkv --impl 4,4,4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/fdiv_kb.txt
python3 tools/kbench.py --libs ab/base.so,ab/fdiv.so --kernels fwd,dq,dkv --impl 4,4,4 --batch 8 --heads 32 --seq 2048 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/fdiv_kb.txt
