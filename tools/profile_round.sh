#!/bin/bash
# Round profile bundle (run on the GPU box through gpurun): PMC passes of the three kernels and the gfx950-corrected HBM
# traffic tied to the kernel source hash FIRST (so that the bench lines of the same run carry roofline.traffic), then the
# bench lines and the rocprofv3 kernel stats of the same bench command.
#   tools/profile_round.sh r03      -> gpurun_out/<tag>_*  (copy what should be judged into profiles/)
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
tools/pmc.sh $OUT/${TAG}_pmc > /dev/null 2>&1 || true
cp $OUT/${TAG}_pmc/summary.txt $OUT/${TAG}_pmc_summary_causal.txt
python3 tools/pmc_traffic.py $OUT/${TAG}_pmc $OUT/${TAG}_pmc_traffic.json > /dev/null
cp $OUT/${TAG}_pmc_traffic.json profiles/pmc_traffic.json    # on the box's scratch copy: bench.py reads it from there
python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
python3 bench.py --config 4 --no-cpu-baseline > $OUT/${TAG}_bench_d128.json 2>> $OUT/${TAG}_bench_default.err
python3 bench.py --layout bshd --no-cpu-baseline --no-config5 > $OUT/${TAG}_bench_bshd.json 2>> $OUT/${TAG}_bench_default.err
cat $OUT/${TAG}_bench_default.json $OUT/${TAG}_bench_d128.json $OUT/${TAG}_bench_bshd.json > $OUT/${TAG}_bench_lines.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python3 bench.py --no-cpu-baseline --no-config5 > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
find $OUT/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats_bench.csv \;
echo done
