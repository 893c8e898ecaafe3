#!/usr/bin/env bash
# Build libqlearn_engine.so for gfx950 in-tree (cross-compiles without a GPU): see Makefile.
# Extra compiler flags (diagnostic builds, e.g. -DQE_STAMPS) go through EXTRA="..." and need OBJ=<other dir>.
set -euo pipefail
cd "$(dirname "$0")"
make -j"${JOBS:-8}" "$@"
echo "built $(pwd)/${LIB:-libqlearn_engine.so}"
