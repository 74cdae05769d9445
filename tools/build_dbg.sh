#!/bin/bash
# Development aid: second build of the libraries with in-kernel s_memrealtime stamps (-DMMG_DEBUG_TIMING)
# into dbglib/ (git-ignored; use with MMGP_LIBDIR=$PWD/dbglib).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/dbglib"
make -C "$root/meshlessmultigridpoisson_amd/csrc" out="$root/dbglib" obj="$root/dbglib/build" \
     HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -DMMG_DEBUG_TIMING $EXTRA" all
