#!/bin/bash
# Collects rocprofv3 PMC counters for one bench.py run, one counter group per pass
# (counters in their own runs, kernel-trace only — see the task's gpurun rules).
# usage: profiles/pmc_passes.sh <out_dir> [bench args...]
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
ls "$OUT"/pass*/*/ | head -20
