#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for
# one bench variant; raw output under gpurun_out/prof_<tag>/, summary printed by
# tools/pmc_summary.py.  Usage: tools/profile_gpu.sh <tag> <bench args...>
# The PMC passes run 16 steps after 8 of warm-up: with the default 8 frames per launch every profiled dispatch of the
# dominant kernel holds 8 frames (tools/make_counters.py divides by that).
set -e -o pipefail
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
# GPU_MAX_HW_QUEUES stays at the box's value (bench.py's batched launches need no extra hardware queues)
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-cpu-baseline --no-extras "$@" > "$out/trace.log" 2>&1
pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/pmc_$name" -- python3 bench.py --steps ${PMC_STEPS:-16} --warmup ${PMC_WARMUP:-8} --no-cpu-baseline --no-extras $BENCH_ARGS > "$out/pmc_$name.log" 2>&1
}
BENCH_ARGS="$*"
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
pass tc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass busy TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_EA0_RDREQ_DRAM_sum
python3 tools/pmc_summary.py "$out" | tee "$out/summary.txt"
