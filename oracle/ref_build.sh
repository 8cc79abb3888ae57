#!/bin/bash
# Builds the REFERENCE's evo_motion_networks sources, where they lie under /root/reference, into
# oracle/_ref/ (git-ignored, never copied into the repo).  Only possible in the authoring container:
# the GPU box has no /root/reference.  Recipe verified in SURVEY.md §8c: link ONLY torch_cpu + c10.
set -e
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
[ -d "$REF/evo_motion_networks/src" ] || { echo "reference not present, skipping"; exit 0; }
T=$(python3 -c 'import torch, os; print(os.path.dirname(torch.__file__))')
mkdir -p "$OUT"
FLAGS="-std=c++20 -O2 -fPIC -I$REF/evo_motion_networks/include -I$T/include -I$T/include/torch/csrc/api/include -D_GLIBCXX_USE_CXX11_ABI=$(python3 -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))')"
if [ ! -f "$OUT/libevo_motion_networks.so" ]; then
  # one object per source, 8 jobs
  SRCS=$(find "$REF/evo_motion_networks/src" -name '*.cpp')
  mkdir -p "$OUT/obj"
  printf '%s\n' $SRCS | xargs -P 8 -I{} sh -c 'o="'"$OUT"'/obj/$(echo {} | md5sum | cut -c1-12).o"; [ -f "$o" ] || g++ '"$FLAGS"' -c {} -o "$o"'
  g++ -shared "$OUT"/obj/*.o -L"$T/lib" -ltorch_cpu -lc10 -Wl,-rpath,"$T/lib" -o "$OUT/libevo_motion_networks.so"
fi
if [ -f "$HERE/ref_golden.cpp" ]; then
  g++ $FLAGS "$HERE/ref_golden.cpp" -L"$OUT" -levo_motion_networks -L"$T/lib" -ltorch_cpu -lc10 \
      -Wl,-rpath,"$OUT" -Wl,-rpath,"$T/lib" -o "$OUT/ref_golden"
fi
if [ -f "$HERE/ref_th.cpp" ]; then
  g++ $FLAGS "$HERE/ref_th.cpp" -L"$OUT" -levo_motion_networks -L"$T/lib" -ltorch_cpu -lc10 \
      -Wl,-rpath,"$OUT" -Wl,-rpath,"$T/lib" -o "$OUT/ref_th"
fi
if [ -f "$HERE/ref_sac.cpp" ]; then
  g++ $FLAGS "$HERE/ref_sac.cpp" -L"$OUT" -levo_motion_networks -L"$T/lib" -ltorch_cpu -lc10 \
      -Wl,-rpath,"$OUT" -Wl,-rpath,"$T/lib" -o "$OUT/ref_sac"
fi
if [ -f "$HERE/ref_loop.cpp" ]; then
  g++ $FLAGS "$HERE/ref_loop.cpp" -L"$OUT" -levo_motion_networks -L"$T/lib" -ltorch_cpu -lc10 \
      -Wl,-rpath,"$OUT" -Wl,-rpath,"$T/lib" -o "$OUT/ref_loop"
fi
if [ -f "$HERE/ref_sac_loop.cpp" ]; then
  g++ $FLAGS "$HERE/ref_sac_loop.cpp" -L"$OUT" -levo_motion_networks -L"$T/lib" -ltorch_cpu -lc10 \
      -Wl,-rpath,"$OUT" -Wl,-rpath,"$T/lib" -o "$OUT/ref_sac_loop"
fi
echo "reference build ok: $OUT"
