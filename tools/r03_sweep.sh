# the reference's own benchmark grid (P:146-167) with this round's kernels and table: four runs, one per (D, dtype)
mkdir -p gpurun_out/r03
for D in 64 128; do for dt in bf16 fp16; do
  python flashattention-from-scratch-with-triton_amd/Performance_Comparison.py $D $dt 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/sweep_B4H8_d${D}_${dt}.txt && tail -3 gpurun_out/r03/sweep_B4H8_d${D}_${dt}.txt
done; done
