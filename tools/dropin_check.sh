#!/bin/bash
# dropin_check.sh — build container only (needs /root/reference): proves that the reference's OWN driver sources compile
# UNMODIFIED against the mirrored headers of alignment-algos_amd/hostcpp and link against libalnhip.so.
# Nothing of the reference is written into the repo: each driver is copied to a temp dir, compiled there, and removed.
#   aa_ali.cpp   the config-1 driver (AASequence, AASubstitutionEval, DPMatrix, Optimal, ConstrainedNearOptimal,
#                FastaOut / PIROut writers, Argv / RCfile parameter plumbing)
# Runtime proof of the same path on a GPU: tests/test_gpu_driver.py (stdout of aaa_hip == stdout of the real aaa).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
REF="${ALN_REFERENCE:-/root/reference}"
if [ ! -d "$REF" ]; then echo "dropin_check: $REF absent (this check runs in the build container only)"; exit 0; fi
make -s -C "$HERE/alignment-algos_amd" "$HERE/alignment-algos_amd/libalnhip.so"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
rc=0
for drv in ${ALN_DROPIN_DRIVERS:-aa_ali.cpp}; do
  cp "$REF/$drv" "$TMP/$drv"
  cmp -s "$REF/$drv" "$TMP/$drv"
  if g++ -std=c++11 -O1 -w -DUNIXVER -I"$HERE/include" -I"$HERE/alignment-algos_amd/hostcpp" "$TMP/$drv" \
        -L"$HERE/alignment-algos_amd" -lalnhip -Wl,-rpath,"$HERE/alignment-algos_amd" -o "$TMP/${drv%.cpp}" 2> "$TMP/err.txt"; then
    echo "dropin_check: $drv (sha256 $(sha256sum "$REF/$drv" | cut -c1-16)) compiles and links unmodified against hostcpp/ + libalnhip.so"
  else
    echo "dropin_check: $drv FAILED"; head -30 "$TMP/err.txt"; rc=1
  fi
done
exit $rc
