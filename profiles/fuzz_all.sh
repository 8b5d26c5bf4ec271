#!/bin/bash
# Every fuzzer of the round, one after the other, one line of summary each (+ the failing seeds):
#   profiles/fuzz_all.sh <outdir>     (on the GPU box; about a minute)
OUT=${1:-gpurun_out/fuzz}
mkdir -p "$OUT"
run() { # $1 = tag, rest = command
  t=$1; shift
  echo "# $*" >> "$OUT/summary.txt"
  "$@" > "$OUT/$t.txt" 2>&1
  grep -v "^seed\|^ " "$OUT/$t.txt" | tail -n 2 >> "$OUT/summary.txt"
  grep "^seed [0-9]* [N:]" "$OUT/$t.txt" | cut -c1-200 >> "$OUT/summary.txt"
}
: > "$OUT/summary.txt"
run plain        python3 profiles/fuzz_oracle.py 1200
run wide         python3 profiles/fuzz_oracle.py 1200 wide
run harsh        python3 profiles/fuzz_oracle.py 600 harsh subnormal
run wide_harsh   python3 profiles/fuzz_oracle.py 400 wide harsh subnormal
run harsh_strict python3 profiles/fuzz_oracle.py 600 harsh
run wharsh_strict python3 profiles/fuzz_oracle.py 400 wide harsh
run short        python3 profiles/fuzz_oracle.py 600 short
run wide_short   python3 profiles/fuzz_oracle.py 300 wide short
run huge         python3 profiles/fuzz_oracle.py 300 huge
run huge_harsh   python3 profiles/fuzz_oracle.py 200 huge harsh subnormal
run huge_short   python3 profiles/fuzz_oracle.py 200 huge short
run tiers        python3 profiles/fuzz_tiers.py 300
run viterbi      python3 profiles/fuzz_viterbi.py 600
run viterbi_h    python3 profiles/fuzz_viterbi.py 600 harsh
run viterbi_huge python3 profiles/fuzz_viterbi.py 200 huge
run train        python3 profiles/fuzz_train.py 300 10
run streams      python3 profiles/fuzz_streams.py 200
run batch        python3 profiles/fuzz_batch.py 300
run options      python3 profiles/fuzz_options.py 200
run threads      python3 profiles/fuzz_threads.py 80
run init         python3 profiles/fuzz_init.py 300
run robust       python3 profiles/fuzz_robust.py 300
run train_huge   python3 profiles/fuzz_train.py 60 6 huge
run init_huge    python3 profiles/fuzz_init.py 60 huge
run train_big    python3 profiles/fuzz_train.py 60 8 big
cat "$OUT/summary.txt"
