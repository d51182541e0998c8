#!/bin/bash
# Builds the library of another commit beside the working tree's: tools/build_baseline.sh [commit] -> brush_amd/csrc/build/
# libbrush_hip_base.so (same ABI; load it with BRUSH_HIP_LIB=... for same-box A/B runs on the GPU).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-HEAD}
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" brush_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/brush_amd/csrc" -j8 ../lib/libbrush_hip.so > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
mkdir -p "$ROOT/brush_amd/csrc/build"
cp "$TMP/brush_amd/lib/libbrush_hip.so" "$ROOT/brush_amd/csrc/build/libbrush_hip_base.so"
rm -rf "$TMP"
echo "built $REV -> brush_amd/csrc/build/libbrush_hip_base.so"
