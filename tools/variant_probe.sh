#!/bin/bash
# Developer A/B on a GPU box (a scratch copy of the repo): run a probe command against libcfx.so and against every build/variants/libcfx_<name>.so
# in turn (each swapped in over compactfusion_amd/libcfx.so; the original is put back afterwards).
#   bash tools/variant_probe.sh <out-file> <command ...>
OUT=$1; shift
LIB=compactfusion_amd/libcfx.so
cp $LIB /tmp/libcfx_main.so
{
echo "== main"; "$@" 2>&1 | grep -v amdgpu.ids
for v in build/variants/libcfx_*.so; do
  [ -f "$v" ] || continue
  echo "== $(basename $v .so | sed s/libcfx_//)"
  cp "$v" $LIB
  "$@" 2>&1 | grep -v amdgpu.ids
done
} > "$OUT"
cp /tmp/libcfx_main.so $LIB
