mkdir -p gpurun_out/r03
L=flashattention-from-scratch-with-triton_amd/libmi355fa.so
python tools/kbench.py --libs $L@1,0,0,$L@2,0,0,$L@4,0,0 --kernels fwd --rounds 3 --reps 3 --batch 64 --seq 8192 --warm-ms 100 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb5_fwd.txt
python tools/kbench.py --libs $L@0,1,0,$L@0,3,0 --kernels dq --rounds 3 --reps 3 --batch 64 --seq 8192 --warm-ms 100 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb5_dq.txt
python tools/kbench.py --libs $L@0,0,2,$L@0,0,3 --kernels dkv --rounds 3 --reps 3 --batch 64 --seq 8192 --warm-ms 100 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb5_dkv.txt
