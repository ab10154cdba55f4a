set -e
for o in "leaf_frac16=0" "leaf_frac16=8" "leaf_frac16=4" "leaf_frac16=0" "leaf_frac16=8" "leaf_frac16=6"; do
  timeout -k 10 200 python -u bench.py --no-cpu-baseline --option $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', d['value'], d['ms_per_step'], d['config'].get('ms_per_frame_one_at_a_time'), d['roofline']['launch']['ms_per_frame'])"
done
for o in "leaf_frac16=0" "leaf_frac16=8"; do
  timeout -k 10 200 python -u bench.py --no-cpu-baseline --workload bunny_1080p --option $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bunny $o', d['value'], d['ms_per_step'], d['config'].get('ms_per_frame_one_at_a_time'), d['roofline']['launch']['ms_per_frame'])"
done
