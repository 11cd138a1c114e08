mkdir -p gpurun_out/r03
timeout -k 10 300 python tools/check_fwd4.py 1 > gpurun_out/r03/check_fwd1.txt 2>&1; tail -1 gpurun_out/r03/check_fwd1.txt; grep FAIL gpurun_out/r03/check_fwd1.txt | head -5
L=flashattention-from-scratch-with-triton_amd/libmi355fa.so
ARMS="$L@1,0,0,$L@4,0,0"
for v in "$@"; do ARMS="$ARMS,ab/$v.so@1,0,0"; done
python tools/kbench.py --libs $ARMS --kernels fwd --rounds 9 --reps 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb7_causal.txt
python tools/kbench.py --libs $ARMS --kernels fwd --rounds 7 --reps 10 --non-causal 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb7_full.txt
python tools/kbench.py --libs $ARMS --kernels fwd --rounds 5 --reps 10 --dim 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb7_d128.txt
