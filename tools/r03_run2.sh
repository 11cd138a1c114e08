set -e
mkdir -p gpurun_out/r03
python tools/check_family.py dkv 2 3 > gpurun_out/r03/check_dkv23.txt 2>&1 || true
tail -2 gpurun_out/r03/check_dkv23.txt
L=flashattention-from-scratch-with-triton_amd/libmi355fa.so
ARMS="$L@0,0,2,$L@0,0,3"
for v in "$@"; do ARMS="$ARMS,ab/$v.so@0,0,3"; done
python tools/kbench.py --libs $ARMS --kernels dkv --rounds 7 --reps 10 --non-causal 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb2_full.txt
python tools/kbench.py --libs $ARMS --kernels dkv --rounds 7 --reps 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb2_causal.txt
