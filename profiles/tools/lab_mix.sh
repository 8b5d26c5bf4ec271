#!/bin/bash
# The statistics kernel (k_mixstats_mfma) with parts of its loads pointed at the stage's first piece
# (lab builds: nop = 2048 posteriors, nox = 4096 frames, nopx = both; RESULTS are wrong by construction).
cd "$GRAFT_REPO_ROOT"
for b in "$@"; do
  lib="$PWD/speech-recognition-hmm-continuous_amd/build/libghmm_lab$b.so"
  [ "$b" = "0" ] && lib="$PWD/speech-recognition-hmm-continuous_amd/libghmm_hip.so"
  for rep in 1 2; do
  GHMM_HIP_LIB="$lib" python3 bench.py --no-extras --no-cpu-baseline --steps 50 --warmup 5 --spinup 100 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('lab $b 10x8 mixstats', k['mixstats'], 'emission', k['emission'], 'step_ms', d['ms_per_step'])"
  done
done
