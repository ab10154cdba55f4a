# A/B of library builds through bench.py's default mode (8 frames per launch, two streams): tools/ab_libs_bench.sh workload tag [tag ...]
w=$1; shift
for round in 1 2; do
for t in "$@"; do
  lib=$PWD/raytracer_amd/lib/librtamd_$t.so; [ "$t" = default ] && lib=$PWD/raytracer_amd/lib/librtamd.so
  RTAMD_DEVICE_LIB=$lib timeout -k 10 300 python -u bench.py --no-cpu-baseline --workload $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', '$w', 'Mrays/s', d['value'], 'ms/frame', d['ms_per_step'], 'one at a time', d['config'].get('ms_per_frame_one_at_a_time'), 'one stream', d['roofline']['launch']['ms_per_frame'])" || exit 1
done
done
