#!/bin/bash
export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# Build an alternative libmi355_attn.so into tools/ab/<name>.so with extra compiler flags for ONE translation unit
# (the other objects are taken from the in-tree build), for same-box A/B runs through MI355_LIB.
# usage: tools/build_variant.sh <name> <file.hip> [extra hipcc flags ...]     (file defaults to prefill_mfma.hip if it does not end in .hip)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=prefill_mfma.hip
case "$1" in *.hip) src=$1; shift;; esac
C=vllm-triton-backend_amd/csrc
mkdir -p tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -DMI355_LAB -Iinclude "$@" -c $C/$src -o tools/ab/$name.o
objs=$(ls $C/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/$name.so $objs tools/ab/$name.o
rm tools/ab/$name.o
echo built tools/ab/$name.so
