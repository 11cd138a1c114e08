#!/bin/bash
# Build one A/B variant of libmi355fa.so into ab/<name>.so (ab/ is git-ignored but travels to the GPU box).
#   tools/build_variant.sh <name> ["-DFOO -DBAR"] [git-rev]     (git-rev: build that revision's csrc instead of the tree)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; REV=$3
SRC=$ROOT/flashattention-from-scratch-with-triton_amd/csrc
TMP=$(mktemp -d /tmp/fa_variant.XXXXXX)
if [ -n "$REV" ]; then
  (cd "$ROOT" && git archive "$REV" flashattention-from-scratch-with-triton_amd/csrc include) | tar -x -C "$TMP"
  SRC=$TMP/flashattention-from-scratch-with-triton_amd/csrc
else
  mkdir -p "$TMP/flashattention-from-scratch-with-triton_amd" "$TMP/include"
  cp -r "$SRC" "$TMP/flashattention-from-scratch-with-triton_amd/csrc"
  cp "$ROOT"/include/*.h "$TMP/include/"
  SRC=$TMP/flashattention-from-scratch-with-triton_amd/csrc
  rm -f "$SRC"/*.o
fi
mkdir -p "$ROOT/ab"
CXX="-O3 -std=c++17 -fno-honor-nans -fno-slp-vectorize -fPIC --offload-arch=gfx950 -Wno-unused-function $FLAGS"
pids=()
for f in "$SRC"/*.hip; do
  /opt/rocm/bin/hipcc $CXX -c "$f" -o "${f%.hip}.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ab/$NAME.so" "$SRC"/*.o
rm -rf "$TMP"
echo "built ab/$NAME.so"
