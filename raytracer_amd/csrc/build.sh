#!/bin/bash
# Builds raytracer_amd/librt_amd.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: every fp64 routine that mirrors reference arithmetic must round each
# product and sum separately; the fp32 sweep spells its FMAs out with __builtin_fmaf.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="${RT_AMD_OUT:-$root/raytracer_amd/librt_amd.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -I"$root/include" -I"$here" -Wall -Wno-unused-value)
objs=()
bdir="${RT_AMD_BUILD_DIR:-$here/build}"
mkdir -p "$bdir"
for src in rt_sweep.hip rt_sweep_apex.hip rt_sweep_cells.hip rt_sort.hip rt_exact.hip rt_transform.hip rt_shade.hip rt_api.hip rt_multi.hip rt_sequence.hip rt_host.cpp rt_obj.cpp; do
  obj="$bdir/${src%.*}.o"
  if [ ! -f "$obj" ] || [ "$here/$src" -nt "$obj" ] || [ "$here/rt_internal.h" -nt "$obj" ] || [ "$here/rt_sweep_common.h" -nt "$obj" ] || [ "$root/include/rt_amd.h" -nt "$obj" ]; then
    echo "hipcc $src"
    extra=()
    # The sweep must stay scalar fp32: SLP-packing it into v_pk_* costs shuffles and doubles the
    # issue time of each packed instruction on gfx950, a net loss for this loop.
    case "$src" in rt_sweep*.hip) extra=(-fno-slp-vectorize --save-temps=obj) ;; esac
    "$HIPCC" "${FLAGS[@]}" "${extra[@]}" ${EXTRA_FLAGS:-} -c "$here/$src" -o "$obj"
  fi
  objs+=("$obj")
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}" -ldl -lpthread -lz
echo "built $out"
