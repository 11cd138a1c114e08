set -e
mkdir -p gpurun_out/r03
python tools/check_family.py dkv 2 3 > gpurun_out/r03/check_dkv23.txt 2>&1 || true
tail -5 gpurun_out/r03/check_dkv23.txt
L=flashattention-from-scratch-with-triton_amd/libmi355fa.so
python tools/kbench.py --libs $L@0,0,2,$L@0,0,3 --kernels dkv --rounds 7 --reps 20 > gpurun_out/r03/kb_dkv_causal.txt 2>&1
cat gpurun_out/r03/kb_dkv_causal.txt
python tools/kbench.py --libs $L@0,0,2,$L@0,0,3 --kernels dkv --rounds 7 --reps 10 --non-causal > gpurun_out/r03/kb_dkv_full.txt 2>&1
cat gpurun_out/r03/kb_dkv_full.txt
