#!/bin/bash
# Runs bench.py once per library variant in rust-raytracer_amd/var/*.so (built with different -D tuning flags).
# Usage: bash tools/tune.sh "<bench args>" [variant names...]
ARGS=$1; shift
VARS=${@:-$(ls rust-raytracer_amd/var/*.so | xargs -n1 basename | sed 's/\.so$//')}
for v in $VARS; do
  out=$(RTCORE_LIB=$PWD/rust-raytracer_amd/var/$v.so timeout -k 10 300 python -u bench.py --no-cpu $ARGS 2>&1 | grep -o '"value": [0-9.]*\|extend_Mrays_per_s": [0-9.]*\|shade_ms_total": [0-9.]*\|other_ms_total": [0-9.]*' | tr '\n' ' ')
  echo "$v: $out"
done
