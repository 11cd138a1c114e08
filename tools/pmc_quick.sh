#!/bin/bash
# Quick PMC comparison of kernel variants: busy cycles, instruction mix and waits for one kbench invocation.
# usage: tools/pmc_quick.sh <outdir> [kbench args...]      (separate passes; never combined with tracing)
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 tools/kbench.py --reps 3 --rounds 2 --warm-ms 50 "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1 || true
cat "$OUT/summary.txt"
