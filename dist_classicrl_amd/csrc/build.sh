#!/usr/bin/env bash
# Build libqlearn_engine.so for gfx950 in-tree (cross-compiles without a GPU).
# -ffp-contract=off: TD arithmetic must round exactly like the reference (qe_device.h, Td<>).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
    -Wall -Wno-unused-function -Wno-unused-result -Wno-pass-failed \
    qe_engine.hip -o libqlearn_engine.so "$@"
echo "built $(pwd)/libqlearn_engine.so"
