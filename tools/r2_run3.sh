set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 500 --timeout-method thread > gpurun_out/r2c_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r2c_tests.log
b() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r2c_bench_$tag.json 2> gpurun_out/r2c_bench_$tag.err || { echo "bench $tag failed"; tail -5 gpurun_out/r2c_bench_$tag.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r2c_bench_$tag.json"))
r=d["roofline"]
print("$tag", d["value"], d["ms_per_step"], "launch", r["launch"], "serial", d["config"].get("ms_per_frame_one_at_a_time"), r["bound"], r["frac"])
PY
}
b dflt --steps 2000 && b s1 --streams 1 --steps 2000 && b steps20 --steps 20 --warmup 5
