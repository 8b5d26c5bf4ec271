#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for b in 0 nostore noexp nomfma; do
  lib="$PWD/speech-recognition-hmm-continuous_amd/build/libghmm_lab$b.so"
  [ "$b" = "0" ] && lib="$PWD/speech-recognition-hmm-continuous_amd/libghmm_hip.so"
  GHMM_HIP_LIB="$lib" python3 bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --mix 64 --utts 12500 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('m64 lab $b', 'emission', k['emission'], 'mixstats', k['mixstats'], 'step_ms', d['ms_per_step'])"
done
