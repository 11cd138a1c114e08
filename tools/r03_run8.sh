# forward family 4: in-tree library against ab/<name>.so variants, both head dims, causal and full; correctness first
mkdir -p gpurun_out/r03
L=flashattention-from-scratch-with-triton_amd/libmi355fa.so
python tools/check_fwd4.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r03/kb8_check.txt
ARMS="$L@1,0,0,$L@4,0,0"
for v in "$@"; do ARMS="$ARMS,ab/$v.so@4,0,0"; done
for D in 64 128; do
python tools/kbench.py --libs $ARMS --kernels fwd --rounds 7 --reps 10 --dim $D 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb8_d${D}_causal.txt
python tools/kbench.py --libs $ARMS --kernels fwd --rounds 7 --reps 10 --dim $D --non-causal 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/kb8_d${D}_full.txt
done
