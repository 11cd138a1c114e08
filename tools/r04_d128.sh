#!/bin/bash
# round 4: the D = 128 backward under the profiler -- PMC passes (busy, instruction mix, waits, LDS) of the three kernels
# and the stamped dK/dV family 2 (the family config 4 runs)
mkdir -p gpurun_out/r04
bash tools/pmc_quick.sh gpurun_out/r04/pmc_d128 --dim 128 --kernels fwd,dq,dkv > gpurun_out/r04/pmc_d128_summary.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/r04/pmc_d128/p4 -- python3 tools/kbench.py --reps 3 --rounds 2 --warm-ms 50 --dim 128 --kernels fwd,dq,dkv > gpurun_out/r04/pmc_d128/p4.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r04/pmc_d128 > gpurun_out/r04/pmc_d128_summary.txt 2>&1
python3 tools/stamps_dkv.py --dim 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/stamps_dkv2_d128.txt
python3 tools/stamps_dkv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/stamps_dkv2_d64.txt
python3 tools/kbench.py --dim 128 --kernels fwd,dq,dkv 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/kb_d128.txt
python3 tools/kbench.py --dim 128 --kernels dq --impl 0,2,0 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04/kb_d128.txt
python3 tools/kbench.py --dim 128 --kernels dkv --impl 0,0,1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04/kb_d128.txt
cat gpurun_out/r04/pmc_d128_summary.txt gpurun_out/r04/stamps_dkv2_d128.txt gpurun_out/r04/kb_d128.txt
