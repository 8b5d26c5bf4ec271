#!/bin/bash
# times one GPU's share of BASELINE configs[3] (64 mixtures) and the 32-mixture shape with the
# product build (or GHMM_HIP_LIB): per-kernel HIP-event averages
cd "$GRAFT_REPO_ROOT"
for mix in 64 32; do
python3 bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --mix $mix --utts 12500 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('mix $mix', 'emission', k['emission'], 'mixstats', k['mixstats'], 'step_ms', d['ms_per_step'])"
done
