#!/bin/bash
# Measurement builds of the library: build/libghmm_lab<tag>.so compiled with extra -D flags
# (csrc/ghmm_mfma.hpp: GHMM_LAB bits switch parts of the emission kernel off, GHMM_EMS_W /
# GHMM_EMS_PF pick its geometry), used through GHMM_HIP_LIB=<path> python bench.py ...
# usage: profiles/tools/lab.sh <tag>:<flags> ...     e.g.  lab.sh nostore:-DGHMM_LAB=2 w12:-DGHMM_EMS_W=12
set -e
HERE=$(cd "$(dirname "$0")/../.." && pwd)
PKG="$HERE/speech-recognition-hmm-continuous_amd"
make -C "$PKG" host >/dev/null
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I"$HERE/include" -Wno-unused-parameter -Wno-pass-failed \
        -munsafe-fp-atomics -ffp-contract=on $flags -c "$PKG/csrc/ghmm_hip.hip" -o "$PKG/build/lab$tag.o" &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$PKG/build/libghmm_lab$tag.so" \
        "$PKG/build/ghmm_synth.o" "$PKG/build/ghmm_io.o" "$PKG/build/ghmm_init.o" "$PKG/build/ghmm_rendezvous.o" "$PKG/build/lab$tag.o" -lm &&
    echo "built lab $tag ($flags)"
  ) &
done
wait
