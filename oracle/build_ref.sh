#!/bin/bash
# Builds oracle/_ref/libref_rela.so from the REAL reference rela/ sources where they lie under
# /root/reference (only runs in the authoring container; the GPU box uses the prebuilt .so, which is
# git-ignored but travels with the gpurun snapshot).  Plain g++ on the reference's own files — the
# reference's cmake build is not used.  Test infrastructure only.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${REFERENCE_ROOT:-/root/reference}
[ -d "$REF/rela" ] || { echo "no reference tree at $REF; keeping prebuilt oracle/_ref"; exit 0; }
OUT="$HERE/_ref"
mkdir -p "$OUT"
if [ "$OUT/libref_rela.so" -nt "$HERE/ref_rela_harness.cc" ] && [ "$OUT/libref_rela.so" -nt "$REF/rela/prioritized_replay.h" ]; then
  exit 0
fi
TORCH=$(python -c "import torch, os; print(os.path.dirname(torch.__file__))")
PYINC=$(python -c "import sysconfig; print(sysconfig.get_paths()['include'])")
g++ -O1 -std=c++17 -fPIC -shared -D_GLIBCXX_USE_CXX11_ABI=1 -w \
  -I"$REF" -I"$TORCH/include" -I"$TORCH/include/torch/csrc/api/include" -I"$PYINC" \
  "$HERE/ref_rela_harness.cc" "$REF/rela/transition.cc" \
  -L"$TORCH/lib" -Wl,-rpath,"$TORCH/lib" -ltorch -ltorch_cpu -lc10 -ltorch_python \
  -o "$OUT/libref_rela.so"
echo "built $OUT/libref_rela.so"
