#!/bin/bash
# Build a variant of libwvhash.so with extra -D flags on ONE source file, for same-session A/B timing:
#   tools/build_variant.sh NAME swt_slide.hip -DWV_SWT_VPRIO=0 ...   ->  tools/_variants/NAME.so
# Use it with WVHASH_LIB=tools/_variants/NAME.so python tools/bench_kernels.py ...
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
CS=$ROOT/image-retrieval-wavelet_amd/csrc
make -C "$CS" -j8 >/dev/null 2>&1
mkdir -p "$ROOT/tools/_variants" "$ROOT/build/variant_obj"
OBJ=$ROOT/build/variant_obj/$NAME.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$CS" "$@" -c -o "$OBJ" "$CS/$SRC" 2>/dev/null
# variants are diagnostic builds: linked with tune_diag.o (the WV_* switches are read from the environment)
OTHERS=$(ls "$ROOT"/build/obj/*.o | grep -v "/$(basename "${SRC%.*}").o" | grep -v "/tune_release.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/_variants/$NAME.so" $OBJ $OTHERS
echo "built tools/_variants/$NAME.so"
