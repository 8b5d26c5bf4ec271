#!/bin/bash
# A/B of measurement builds in one session: bench emission / step time (3 rounds), the 64-mixture
# share, the 2 000-state emission and the decode timings.  usage: lab_ab.sh <tag> ...
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for b in "$@"; do
  lib="$PWD/speech-recognition-hmm-continuous_amd/build/libghmm_lab$b.so"
  GHMM_HIP_LIB="$lib" python3 bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 5 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('lab $b', 'emission', k['emission'], 'step_ms', d['ms_per_step'])"
done; done
for b in "$@"; do
  lib="$PWD/speech-recognition-hmm-continuous_amd/build/libghmm_lab$b.so"
  echo "== $b"
  GHMM_HIP_LIB="$lib" bash profiles/tools/m64_time.sh
  GHMM_HIP_LIB="$lib" python3 profiles/config5_emission.py | tail -n 1
  GHMM_HIP_LIB="$lib" python3 profiles/decode_time.py
done
