#!/bin/bash
# Build A/B variants of the library: build_variants.sh name1 "flags1" name2 "flags2" ...
# -> challenge_amd/csrc/libiris_frontend_<name>.so (git-ignored; they travel to the GPU box)
cd "$(dirname "$0")/../challenge_amd/csrc"
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $2 -shared -o libiris_frontend_$1.so iris_frontend.hip &
  shift 2
done
wait
ls -la libiris_frontend_*.so
