#!/bin/bash
# Parts of the emission kernels, 10x8 (bench) and 64 mixtures (one GPU's share of configs[3]), by
# builds with parts compiled out (profiles/tools/lab.sh noexp:-DGHMM_LAB=1 nostore:-DGHMM_LAB=2
# nomfma:-DGHMM_LAB=4); tag 0 = the product.  The lab builds' RESULTS are wrong by construction.
cd "$GRAFT_REPO_ROOT"
for b in "$@"; do
  lib="$PWD/speech-recognition-hmm-continuous_amd/build/libghmm_lab$b.so"
  [ "$b" = "0" ] && lib="$PWD/speech-recognition-hmm-continuous_amd/libghmm_hip.so"
  for rep in 1 2; do
  GHMM_HIP_LIB="$lib" python3 bench.py --no-extras --no-cpu-baseline --steps 50 --warmup 5 --spinup 100 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('lab $b 10x8 emission', k['emission'], 'step_ms', d['ms_per_step'])"
  done
  GHMM_HIP_LIB="$lib" python3 bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --spinup 0 --mix 64 --utts 12500 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('lab $b m64 emission', k['emission'], 'mixstats', k['mixstats'], 'step_ms', d['ms_per_step'])"
done
