#!/bin/sh
# build_ref.sh — TEST INFRASTRUCTURE.  Compiles the REAL reference from the
# sources where they lie under /root/reference into oracle/_ref/ (git-ignored).
# Nothing of the reference is copied into the repo: the raised-capacity builds
# pipe the source through sed into a scratch directory that is deleted on exit.
#
# Products (all in oracle/_ref/):
#   hmm-continuous-train-fs            trainer, diagonal, as shipped (9-d, 3 mix)
#   recognition-continuous-test-fs     recogniser, diagonal, as shipped
#   hmm-continuous-train-fs-big        same trainer, MAX_COEF 39 / MAX_MIX 64 / P 1
#   recognition-continuous-test-fs-big same recogniser, raised the same way
#   ref_harness                        function-level dumper (oracle/ref_harness.c)
#
# The reference's own Makefiles are not used (broken paths, -pg).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${GHMM_REFERENCE:-/root/reference}
if [ ! -d "$REF/train/source/hmm-fs" ]; then
    echo "build_ref.sh: $REF not present — skipping (GPU box uses the prebuilt oracle/_ref)"
    exit 0
fi
OUT="$HERE/_ref"
mkdir -p "$OUT"
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
TF="$REF/train/source/hmm-fs/hmm_continuous_fs.c"
RF="$REF/test/source/recognition-fs/recognition_continuous_fs.c"
CC="${CC:-gcc} -O2 -w -ffp-contract=off"

raise() { # $1 = source
    sed -E \
        -e 's/^(#define[ \t]+MAX_COEF_NUMBER)[ \t]+[0-9]+/\1 39/' \
        -e 's/^(#define[ \t]+MAX_MIXTURE_NUMBER)[ \t]+[0-9]+/\1 64/' \
        -e 's/^(#define[ \t]+MAX_STATES_NUMBER)[ \t]+[0-9]+/\1 20/' \
        -e 's/^(#define[ \t]+MAX_PARAMETERS_NUMBER)[ \t]+[0-9]+/\1 1/' \
        "$1"
}

$CC "$TF" -o "$OUT/hmm-continuous-train-fs" -lm
$CC "$RF" -o "$OUT/recognition-continuous-test-fs" -lm
raise "$TF" > "$TMP/tf_big.c"
raise "$RF" > "$TMP/rf_big.c"
$CC "$TMP/tf_big.c" -o "$OUT/hmm-continuous-train-fs-big" -lm
$CC "$TMP/rf_big.c" -o "$OUT/recognition-continuous-test-fs-big" -lm
$CC -DREF_TF_SOURCE="\"$TMP/tf_big.c\"" "$HERE/ref_harness.c" -o "$OUT/ref_harness" -lm
echo "build_ref.sh: built $(ls "$OUT" | tr '\n' ' ')"
