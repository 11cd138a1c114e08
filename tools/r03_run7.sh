# in-tree library (arm 1) against ab/<name>.so variants (further arms): bit-identity of dK/dV family 3 vs 2 first, then timing
mkdir -p gpurun_out/r03
L=flashattention-from-scratch-with-triton_amd/libmi355fa.so
python tools/check_family.py dkv 2 3 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r03/kb7_check.txt
ARMS="$L@0,0,3"
for v in "$@"; do ARMS="$ARMS,ab/$v.so@0,0,3"; done
python tools/kbench.py --libs $ARMS --kernels dkv --rounds 7 --reps 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb7_causal.txt
python tools/kbench.py --libs $ARMS --kernels dkv --rounds 7 --reps 10 --non-causal 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb7_full.txt
