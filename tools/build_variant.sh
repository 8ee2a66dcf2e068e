#!/bin/bash
# Build an alternative libmi355_attn.so into tools/ab/<name>.so with extra compiler flags for prefill_mfma.hip only
# (the other objects are taken from the in-tree build), for same-box A/B runs through MI355_LIB.
# usage: tools/build_variant.sh <name> [extra hipcc flags for prefill_mfma.hip ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
C=vllm-triton-backend_amd/csrc
mkdir -p tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Iinclude "$@" -c $C/prefill_mfma.hip -o tools/ab/$name.o
objs=$(ls $C/build/*.o | grep -v prefill_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/$name.so $objs tools/ab/$name.o
rm tools/ab/$name.o
echo built tools/ab/$name.so
