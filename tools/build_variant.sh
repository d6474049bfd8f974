#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> [-Dflags...]   -> build/libspair_<name>.so with that one source rebuilt with the flags
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p build/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -Ispair_pytorch_amd/csrc -Wno-unused-result -Wno-pass-failed "$@" -c spair_pytorch_amd/csrc/$src -o build/var_$name/${src%.hip}.o
objs=""
for f in spair_pytorch_amd/csrc/build/*.o; do b=$(basename $f); if [ "$b" == "${src%.hip}.o" ]; then objs="$objs build/var_$name/$b"; else objs="$objs $f"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/libspair_$name.so $objs
echo built build/libspair_$name.so
