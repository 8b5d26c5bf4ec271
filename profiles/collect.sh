#!/bin/bash
# Round profile collection on the GPU box (everything lands under gpurun_out/<tag>/):
#   kernel-trace statistics of the bench workload, the four PMC passes, and kernel-trace
#   statistics of the decode, 64-mixture and 2 000-state side workloads.
# usage: profiles/collect.sh <tag>
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
stats() { # $1 = name, rest = command after --
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$n" -- "$@" > "$OUT/$n.log" 2>&1 || echo "$n failed"
  f=$(ls "$OUT/$n"/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${n}_kernel_stats.csv"
}
stats bench python3 bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 3
echo "bench stats done"
bash profiles/pmc_passes.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1
python3 profiles/pmc_summary.py "$OUT/pmc" k_emission_sched k_mixstats_mfma k_scan_combine k_scan_pair k_combine k_backward_fix k_reduce_all k_mstep_mfma > "$OUT/pmc_summary.txt"
cp profiles/emission_traffic.json "$OUT/emission_traffic_before.json"
python3 profiles/emission_traffic.py "$OUT/pmc" "$TAG" > "$OUT/emission_traffic.log" 2>&1 && cp profiles/emission_traffic.json "$OUT/emission_traffic.json"
echo "pmc done"
stats config5 python3 profiles/config5_emission.py
echo "config5 done"
stats decode python3 profiles/decode_time.py
echo "decode done"
stats m64 python3 bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --mix 64 --utts 12500
echo "m64 done"
stats refinit python3 profiles/train_trace.py
echo "refinit done"
for f in "$OUT"/*.log; do echo "== $f"; tail -n 3 "$f"; done
