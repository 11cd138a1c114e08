set -e
mkdir -p gpurun_out/r03
L=flashattention-from-scratch-with-triton_amd/libmi355fa.so
ARMS="$L@1,0,0,$L@4,0,0"
for v in "$@"; do ARMS="$ARMS,ab/$v.so@4,0,0"; done
python tools/kbench.py --libs $ARMS --kernels fwd --rounds 7 --reps 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb3_causal.txt
python tools/kbench.py --libs $ARMS --kernels fwd --rounds 7 --reps 10 --non-causal 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb3_full.txt
python tools/kbench.py --libs $L@1,0,0,$L@4,0,0 --kernels fwd --rounds 5 --reps 10 --dim 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb3_d128_causal.txt
python tools/kbench.py --libs $L@1,0,0,$L@4,0,0 --kernels fwd --rounds 5 --reps 10 --dim 128 --non-causal 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb3_d128_full.txt
