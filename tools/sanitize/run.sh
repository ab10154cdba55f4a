#!/bin/bash
# CPU only.  Builds the host layer from source with sanitizers and runs tools/sanitize/host_sanitize.cpp.
#   tools/sanitize/run.sh asan   AddressSanitizer + UndefinedBehaviorSanitizer
#   tools/sanitize/run.sh tsan   ThreadSanitizer (the parallel BVH builder)
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$here/../..
mode=${1:-asan}; out=${TMPDIR:-/tmp}/rtamd_host_$mode
src="$root/raytracer_amd/csrc/host/model.cpp $root/raytracer_amd/csrc/host/bvh.cpp $root/raytracer_amd/csrc/host/resource_parser.cpp $root/raytracer_amd/csrc/host/camera.cpp $root/raytracer_amd/csrc/host/host_capi.cpp"
case $mode in
  asan) san="-fsanitize=address,undefined -fno-sanitize-recover=undefined" ;;
  tsan) san="-fsanitize=thread" ;;
  *) echo "usage: $0 asan|tsan"; exit 2 ;;
esac
g++ -O1 -g -std=c++17 -ffp-contract=off -fno-omit-frame-pointer $san -o "$out" "$here/host_sanitize.cpp" $src -lpthread
shift || true
"$out" "$@"
