#!/bin/bash
# GPU box.  Host code of the device library under ASan + UBSan (device code is not instrumented).
# Build here (hipcc cross-compiles): tools/sanitize/run_device.sh build ; run on the box: tools/sanitize/run_device.sh run
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$here/../..
bin=$root/tools/micro/bin/device_host_sanitize
case ${1:-build} in
  build)
    mkdir -p "$root/tools/micro/bin"
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
      -fgpu-flush-denormals-to-zero -fno-gpu-rdc -fno-slp-vectorize -Wno-unused-function -Wno-unused-result \
      -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer \
      -o "$bin" "$here/device_host_sanitize.cpp" "$root/raytracer_amd/csrc/device/rtamd_api.hip" "$root/raytracer_amd/csrc/device/rtamd_multi.cpp" -ldl ;;
  run) ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 "$bin" ;;
esac
