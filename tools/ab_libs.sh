# A/B of library builds (raytracer_amd/lib/librtamd_<tag>.so) on one box: tools/ab_libs.sh workload "arms" tag [tag ...]
w=$1; arms=$2; shift 2
for round in 1 2; do
for t in "$@"; do
  lib=$PWD/raytracer_amd/lib/librtamd_$t.so; [ "$t" = default ] && lib=$PWD/raytracer_amd/lib/librtamd.so
  RTAMD_DEVICE_LIB=$lib timeout -k 10 300 python -u tools/kbench.py --workload $w --rounds 8 $arms 2>/dev/null | sed "s/^/$t /" || exit 1
done
done
