mkdir -p gpurun_out/r03
L=flashattention-from-scratch-with-triton_amd/libmi355fa.so
ARMS="$L@0,0,3"
for v in "$@"; do ARMS="$ARMS,ab/$v.so@0,0,3"; done
python tools/kbench.py --libs $ARMS --kernels dkv --rounds 7 --reps 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb6_causal.txt
python tools/kbench.py --libs $ARMS --kernels dkv --rounds 7 --reps 10 --non-causal 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb6_full.txt
python tools/check_family.py dkv 2 3 --lib ab/$1.so 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r03/kb6_check.txt
