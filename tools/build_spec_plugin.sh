#!/bin/bash
# build_spec_plugin.sh <cascade.xml|.hcc> <out.so> [stages=8]
# Compiled early stages for a cascade the library was not built with (include/hip_cascade.h hc_spec_plugin_load):
# packs the XML if needed, generates the stage code and compiles it for gfx950 into a shared object.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CASC=$1; OUT=$2; NST=${3:-8}
[ -n "$CASC" ] && [ -n "$OUT" ] || { echo "usage: $0 <cascade.xml|.hcc> <out.so> [stages]" >&2; exit 64; }
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
case "$CASC" in
  *.hcc) HCC=$CASC ;;
  *) HCC=$TMP/cascade.hcc; python3 "$ROOT/tools/pack_cascade.py" "$CASC" "$HCC" ;;
esac
python3 "$ROOT/tools/gen_spec.py" "$HCC" plugin "$NST" "$TMP/spec_plugin.inc" > /dev/null
${HIPCC:-/opt/rocm/bin/hipcc} -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden --offload-arch=gfx950 \
  -I "$ROOT/ugproj-detector_amd/csrc" -DHC_SPEC_FILE="\"$TMP/spec_plugin.inc\"" -x hip -shared \
  -o "$OUT" "$ROOT/ugproj-detector_amd/csrc/spec_plugin.hip"
echo "wrote $OUT"
