#!/bin/bash
# Build the standalone Winograd microbenchmark library from the headers of a git revision (or of the working tree: WORK), as
# scripts/microbench/libwino_<tag>.so - for A/B timing of two kernel versions in ONE gpurun session (scripts/gpu_wino_ab.sh).
# usage: scripts/build_wino_ab.sh <tag> <rev|WORK> [extra hipcc flags]
set -e
tag=$1; rev=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/wino_ab.XXXXXX)
mkdir -p $tmp/challenge_amd/csrc $tmp/scripts/microbench $tmp/include
if [ "$rev" = WORK ]; then
  cp $root/challenge_amd/csrc/*.h $tmp/challenge_amd/csrc/; cp $root/scripts/microbench/wino_conv.hip $tmp/scripts/microbench/; cp $root/include/*.h $tmp/include/
else
  for f in $(git -C $root ls-tree --name-only $rev challenge_amd/csrc/ | grep '\.h$'); do git -C $root show $rev:$f > $tmp/$f; done
  git -C $root show $rev:scripts/microbench/wino_conv.hip > $tmp/scripts/microbench/wino_conv.hip
  for f in $(git -C $root ls-tree --name-only $rev include/); do git -C $root show $rev:$f > $tmp/$f; done
fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared "$@" -o $root/scripts/microbench/libwino_$tag.so $tmp/scripts/microbench/wino_conv.hip 2> $tmp/err.log || { grep -A6 "error" $tmp/err.log | head -40; rm -rf $tmp; exit 1; }
rm -rf $tmp
ls -la $root/scripts/microbench/libwino_$tag.so
