#!/bin/bash
# Round-3 evidence run (same steps as tools/r2_profile.sh) on the GPU box: rocprofv3 kernel-trace + PMC passes of the bench (non-overlapping batch launches),
# counter records for bench.py, then the bench lines themselves.  Usage: tools/r3_profile.sh [tag]   (default r03)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
tag=${1:-r03}
tools/profile_gpu.sh ${tag}_atrium --streams 1 --batch 8 > gpurun_out/${tag}_atrium_profile.log 2>&1 || { echo "atrium profile failed"; tail -5 gpurun_out/${tag}_atrium_profile.log; exit 1; }
cp gpurun_out/prof_${tag}_atrium/summary.txt gpurun_out/${tag}_atrium_summary.txt
python3 tools/make_counters.py gpurun_out/${tag}_atrium_summary.txt atrium_1080p 8 valu_issue profiles/${tag}_atrium_summary.txt || exit 1
cp profiles/counters_atrium_1080p.json gpurun_out/
find gpurun_out/prof_${tag}_atrium/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_atrium_kernel_stats.csv \;
PMC_STEPS=4 PMC_WARMUP=2 tools/profile_gpu.sh ${tag}_soup10m --workload soup_10m_4k --streams 1 --batch 8 --steps 40 --warmup 8 > gpurun_out/${tag}_soup10m_profile.log 2>&1 || { echo "soup profile failed"; tail -5 gpurun_out/${tag}_soup10m_profile.log; exit 1; }
cp gpurun_out/prof_${tag}_soup10m/summary.txt gpurun_out/${tag}_soup10m_summary.txt
python3 tools/make_counters.py gpurun_out/${tag}_soup10m_summary.txt soup_10m_4k 8 tcp_access_rate profiles/${tag}_soup10m_summary.txt || exit 1
cp profiles/counters_soup_10m_4k.json gpurun_out/
find gpurun_out/prof_${tag}_soup10m/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_soup10m_kernel_stats.csv \;
rm -rf gpurun_out/prof_${tag}_atrium/pmc_* gpurun_out/prof_${tag}_soup10m/pmc_* gpurun_out/prof_${tag}_atrium/trace gpurun_out/prof_${tag}_soup10m/trace
timeout -k 10 300 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 python3 bench.py --streams 1 > gpurun_out/${tag}_bench_streams1.json 2>> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 python3 bench.py --batch 1 --streams 1 > gpurun_out/${tag}_bench_serial.json 2>> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_20steps.json 2>> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 400 python3 bench.py --workload soup_10m_4k --steps 100 --warmup 10 > gpurun_out/${tag}_bench_c4_soup10m_4k.json 2>> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 python3 bench.py --workload bunny_1080p > gpurun_out/${tag}_bench_c2_bunny_1080p.json 2>> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 python3 bench.py --workload f16_640x480 > gpurun_out/${tag}_bench_c1_f16_640x480.json 2>> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 400 python3 tools/bench_c5.py --spp 64 > gpurun_out/${tag}_c5_multibounce_4k_64spp.json 2>> gpurun_out/${tag}_bench.err || exit 1
for f in bench bench_streams1 bench_serial bench_20steps bench_c4_soup10m_4k bench_c2_bunny_1080p bench_c1_f16_640x480 c5_multibounce_4k_64spp; do python3 - <<PY
import json
d=json.load(open("gpurun_out/${tag}_$f.json")); r=d.get("roofline",{})
print("$f", d["value"], d.get("ms_per_step"), r.get("bound"), r.get("frac"), (r.get("hbm") or {}).get("frac"), (r.get("launch") or {}).get("ms_per_frame"))
PY
done
