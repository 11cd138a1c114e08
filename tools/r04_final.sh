#!/bin/bash
# Round 4 evidence bundle (run on the GPU box through gpurun; ab/stamps.so = a -DFA_STAMPS build of the current tree):
#   stamps of the three family-4 kernels, repeat-launch race stress of every schedule family, the reference's benchmark
#   grid (B4 H8) per head dim and dtype, the small-shape step times, then tools/profile_round.sh (PMC, traffic, bench lines,
#   rocprofv3 kernel stats).  Everything lands in gpurun_out/r04f/ and gpurun_out/r04_*; copy what is judged to profiles/.
O=gpurun_out/r04f
mkdir -p $O
{
  python3 tools/stamps_fwd4.py ab/stamps.so
  python3 tools/stamps_fwd4.py --dim 128 ab/stamps.so
} 2>&1 | grep -v amdgpu.ids > $O/stamps_fwd4.txt
python3 tools/stamps_dq4.py ab/stamps.so 2>&1 | grep -v amdgpu.ids > $O/stamps_dq4_causal.txt
python3 tools/stamps_dq4.py --non-causal ab/stamps.so 2>&1 | grep -v amdgpu.ids > $O/stamps_dq4_noncausal.txt
python3 tools/stamps_dq4.py --dkv ab/stamps.so 2>&1 | grep -v amdgpu.ids > $O/stamps_dkv4_causal.txt
python3 tools/stamps_dq4.py --dkv --non-causal ab/stamps.so 2>&1 | grep -v amdgpu.ids > $O/stamps_dkv4_noncausal.txt
python3 tools/stamps_dkv.py --dim 128 2>&1 | grep -v amdgpu.ids > $O/stamps_dkv2_d128.txt
echo stamps done
{
  for k in fwd dq dkv; do for f in 1 2 3 4; do
    timeout -k 10 200 python3 tools/race_stress.py $k $f --runs 150 --poison | grep -v "does not take"
  done; done
  for k in fwd dq dkv; do for f in 1 2 4; do
    timeout -k 10 200 python3 tools/race_stress.py $k $f --runs 80 --poison --dim 128 | grep -v "does not take"
  done; done
  timeout -k 10 200 python3 tools/race_stress.py dkv 4 --runs 1000
  timeout -k 10 200 python3 tools/race_stress.py dkv 4 --runs 300 --shape 8,32,2048
  timeout -k 10 200 python3 tools/race_stress.py dq 4 --runs 300 --shape 1,16,16384
  timeout -k 10 200 python3 tools/race_stress.py fwd 4 --runs 300 --shape 2,16,8192
} 2>&1 | grep -v amdgpu.ids > $O/race_stress.txt
echo race done
for D in 64 128; do for dt in bf16 fp16; do
  python flashattention-from-scratch-with-triton_amd/Performance_Comparison.py $D $dt 2>&1 | grep -v amdgpu.ids > $O/sweep_B4H8_d${D}_${dt}.txt
done; done
echo sweep done
python3 tools/small_shapes.py 2>&1 | grep -v amdgpu.ids > $O/small_shapes.txt
{ for S in 512 1024 2048; do python3 tools/graph_step.py 4 8 $S 64; python3 tools/graph_step.py 4 8 $S 64 fp16; done; } 2>&1 | grep -v amdgpu.ids > $O/graph_step_small.txt
echo small done
bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1
echo bundle done
