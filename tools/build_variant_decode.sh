#!/bin/bash
export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# As tools/build_variant.sh, for decode_splitkv.hip: tools/build_variant_decode.sh <name> [extra hipcc flags ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
C=vllm-triton-backend_amd/csrc
mkdir -p tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -DMI355_LAB -Iinclude "$@" -c $C/decode_splitkv.hip -o tools/ab/$name.o
objs=$(ls $C/build/*.o | grep -v decode_splitkv.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/$name.so $objs tools/ab/$name.o
rm tools/ab/$name.o
echo built tools/ab/$name.so
