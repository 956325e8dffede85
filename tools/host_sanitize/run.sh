#!/bin/bash
# Host-only check: the scene builder (csrc/scene_build.cpp: threaded sorts, per-mesh builds, cost-table collapse,
# rebuilt interior) under the address and undefined-behaviour sanitizers of g++.  No GPU code is involved.
# Usage: bash tools/host_sanitize/run.sh     (from the repository root; takes a few minutes)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/rtcore_host_sanitize
mkdir -p "$OUT"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -static-libasan -static-libubsan -fno-omit-frame-pointer \
    -ffp-contract=off -mavx2 -mfma -I"$ROOT/rust-raytracer_amd/csrc" -I"$ROOT/include" \
    "$ROOT/tools/host_sanitize/build_harness.cpp" "$ROOT/rust-raytracer_amd/csrc/scene_build.cpp" -o "$OUT/harness" -lpthread
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 "$OUT/harness"
# the scene-file / OBJ / RGBE readers on whole, truncated and corrupted files (pass files as arguments;
# by default the reference's example scenes and one tea-set model when that checkout is present)
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -static-libasan -static-libubsan -fno-omit-frame-pointer \
    -I"$ROOT/include" -I"$ROOT/tools/host_sanitize" "$ROOT/tools/host_sanitize/loader_harness.cpp" -o "$OUT/loader_harness" -lz
FILES=("$@")
if [ ${#FILES[@]} -eq 0 ] && [ -d /root/reference/scene ]; then
    FILES=(/root/reference/scene/example/*.ron /root/reference/scene/models/newell_teaset/teacup.obj "$ROOT"/tests/golden/sky_512x256_rle.hdr)
fi
if [ ${#FILES[@]} -gt 0 ]; then
    ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 "$OUT/loader_harness" "$OUT" "${FILES[@]}"
fi
