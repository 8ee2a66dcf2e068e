#!/bin/bash
export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# Build and run the HBM streaming-read reference point (tools/probes/hbm_read.hip); on a GPU box:
#   tools/hbm_reference_point.sh > profiles/rNN/hbm_read_reference_point.log
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab
[ -x tools/ab/hbm_read ] && [ tools/ab/hbm_read -nt tools/probes/hbm_read.hip ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/hbm_read.hip -o tools/ab/hbm_read
exec tools/ab/hbm_read
