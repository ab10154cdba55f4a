#!/bin/bash
# A/B of library builds on the GPU box: tools/ab.sh out_file lib1.so lib2.so ... (libs under raytracer_amd/lib/exp/)
# Per library and workload one process (tools/kbench.py); serial kernel time and 4 frames in flight.
out=$1; shift
: > "$out"
for wl in ${AB_WORKLOADS:-atrium_1080p soup_1m_4k}; do
  for lib in "$@"; do
    echo "== $wl $lib serial" >> "$out"
    RTAMD_DEVICE_LIB=$PWD/raytracer_amd/lib/exp/$lib timeout -k 10 240 python tools/kbench.py --workload $wl --rounds ${AB_ROUNDS:-12} fused trace:mode=trace inc:mode=incoherent >> "$out" 2>&1 || exit 1
    echo "== $wl $lib inflight4" >> "$out"
    RTAMD_DEVICE_LIB=$PWD/raytracer_amd/lib/exp/$lib timeout -k 10 240 python tools/kbench.py --workload $wl --rounds ${AB_ROUNDS:-12} --inflight 4 fused >> "$out" 2>&1 || exit 1
  done
done
