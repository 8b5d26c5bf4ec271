#!/bin/bash
# 64-mixture training share and the 2 000-state emission with measurement builds (tags; 0 = product)
cd "$GRAFT_REPO_ROOT"
for b in "$@"; do
  lib="$PWD/speech-recognition-hmm-continuous_amd/build/libghmm_lab$b.so"
  [ "$b" = "0" ] && lib="$PWD/speech-recognition-hmm-continuous_amd/libghmm_hip.so"
  echo "== $b"
  GHMM_HIP_LIB="$lib" bash profiles/tools/m64_time.sh
  GHMM_HIP_LIB="$lib" python3 profiles/config5_emission.py | tail -n 1
done
