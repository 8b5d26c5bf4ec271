#!/bin/bash
# Times the emission kernel of measurement builds (profiles/tools/lab.sh) with bench.py's
# HIP-event pass; tags listed in $CHECK also run the oracle parity tests on that build.
# usage (GPU box): [CHECK="tag ..."] profiles/tools/lab_run.sh <tag> ...   (tag 0 = the product build)
HERE=$(cd "$(dirname "$0")/../.." && pwd)
cd "$HERE"
for b in "$@"; do
  lib="$HERE/speech-recognition-hmm-continuous_amd/build/libghmm_lab$b.so"
  [ "$b" = "0" ] && lib="$HERE/speech-recognition-hmm-continuous_amd/libghmm_hip.so"
  GHMM_HIP_LIB="$lib" python3 bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('lab $b', 'emission', k['emission'], 'mixstats', k['mixstats'], 'fwd', k['forward'], 'bwd', k['backward'], 'step_ms', d['ms_per_step'])"
done
for b in $CHECK; do
  lib="$HERE/speech-recognition-hmm-continuous_amd/build/libghmm_lab$b.so"
  echo "parity on lab $b:"
  GHMM_HIP_LIB="$lib" python3 -m pytest tests/test_gpu_parity.py -q -x -k "estep_against_oracle or rows_against or both_kernel_tiers or viterbi_paths or fuzz_estep" 2>&1 | tail -3
done
