#!/bin/bash
# build_variant.sh NAME SOURCE.hip [extra hipcc flags]: a copy of libqplace.so with one translation unit rebuilt with
# extra flags (development: instrumented kernels, launch knobs), as tools/bin/libqplace_NAME.so; use it with QPLACE_LIB=...
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
csrc="$root/quantum-simulator_amd/csrc"
mkdir -p "$root/tools/bin"
obj="$root/tools/bin/${name}_$(basename "$src" .hip).o"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c "$csrc/$src" -o "$obj"
others=$(ls "$csrc"/build/*.o | grep -v "/$(basename "$src" .hip).o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $others "$obj" -o "$root/tools/bin/libqplace_${name}.so"
rm -f "$obj"
echo "$root/tools/bin/libqplace_${name}.so"
