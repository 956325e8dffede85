#!/bin/bash
# Builds rust-raytracer_amd/var/<name>.so from the current sources with extra compiler flags (tuning experiments;
# tools/tune.sh runs bench.py once per variant, RTCORE_LIB selects one for a test run).
# Usage: bash tools/build_variant.sh <name> "<extra flags>"
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../rust-raytracer_amd"
mkdir -p var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mavx2 -mfma -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -fno-slp-vectorize $FLAGS \
  -shared -o var/$NAME.so csrc/rtcore.hip csrc/post.hip csrc/multi.hip csrc/gpu_build.hip csrc/compact.hip csrc/scene_build.cpp -pthread -ldl 2>&1 | grep -E "error" || true
echo "built var/$NAME.so ($FLAGS)"
