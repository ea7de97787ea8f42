#!/bin/bash
# The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; GPU sanitizers are not available on this pool):
# builds oracle/gme_oracle.c into /tmp with -fsanitize=address,undefined and runs the oracle tests that go through it --
# goldens, corner cases and the random cross-check against the NumPy oracle.   usage: bash tools/oracle_sanitize.sh
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -shared -o /tmp/libgme_oracle_asan.so oracle/gme_oracle.c -lm
cat > /tmp/run_oracle_asan.py <<PY
import sys
sys.path[:0] = ["$PWD/tests", "$PWD"]
import helpers
helpers.ORACLE_SO = "/tmp/libgme_oracle_asan.so"
import pytest
sys.exit(pytest.main(["$PWD/tests/test_oracle.py", "-x", "-q", "-p", "no:cacheprovider", "-k",
                      "c_oracle_small or random_frames or corner_cases or degenerate or compensate or affine_fields or float32_order or first_parameters or gme_stages"]))
PY
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python3 /tmp/run_oracle_asan.py
