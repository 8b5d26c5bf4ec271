#!/bin/bash
# rocprofv3 kernel statistics of the bench workload with a measurement build (tag; 0 = product)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for b in "$@"; do
  lib="$PWD/speech-recognition-hmm-continuous_amd/build/libghmm_lab$b.so"
  [ "$b" = "0" ] && lib="$PWD/speech-recognition-hmm-continuous_amd/libghmm_hip.so"
  out=gpurun_out/labstats_$b
  rm -rf "$out"; mkdir -p "$out"
  GHMM_HIP_LIB="$lib" rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 3 > "$out/log.txt" 2>&1
  echo "== $b: $(grep -o '"ms_per_step": [0-9.]*' "$out/log.txt")"
  python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
tot = 0
for r in list(csv.reader(open(f)))[1:8]:
    print("  ", r[0].replace("void ghmm::", "").replace("ghmm::", "")[:36].ljust(36), r[1].rjust(5), f"{float(r[3]) / 1000:8.2f} us")
    tot += float(r[3]) / 1000 if int(r[1]) > 400 else 0
print("   sum of the per-iteration kernels", round(tot, 2), "us")
PY
done
