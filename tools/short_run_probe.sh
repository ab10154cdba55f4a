cd "${GRAFT_REPO_ROOT:-.}"
for b in 4 5 7 8 0; do for rep in 1 2 3; do timeout -k 5 120 python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('batch', $b, d['config']['frames_per_launch'], d['ms_per_step'], d['value'])"; done; done
