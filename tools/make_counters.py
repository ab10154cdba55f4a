#!/usr/bin/env python3
"""profiles/counters_<workload>.json from a tools/profile_gpu.sh summary: the per-FRAME counter figures bench.py puts
into its roofline record (HBM traffic, VALU wave-instructions, L1 look-ups) and the roof that bounds the workload.
Usage: tools/make_counters.py <summary.txt> <workload> <frames per launch of the profiled run> <valu_issue|tcp_access_rate> [kept copy of the summary]"""
import json, os, re, sys
summary, workload, frames, bound = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
kept = sys.argv[5] if len(sys.argv) > 5 else summary
vals, cur, busy = {}, None, {}
for line in open(summary):
    m = re.match(r"\[(pmc_\w+)\] (.*)", line)
    if m:
        # the dominant kernel: fused, no test counters, the instantiation batches use (last template argument false)
        cur = m.group(1) if re.search(r"k_persistent<0, false, (true|false), (true|false), false>", m.group(2)) else None
        continue
    m = re.match(r"\s+(\w+)\s+([\d.]+)\s+\(n=(\d+)\)", line)
    if m and cur:
        vals[m.group(1)] = (float(m.group(2)), int(m.group(3)))
        if cur == "pmc_busy":
            busy[m.group(1)] = float(m.group(2))      # GRBM_GUI_ACTIVE of the SAME pass as TA_BUSY
need = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "TCP_TOTAL_CACHE_ACCESSES_sum")
missing = [k for k in need if k not in vals]
if missing:
    raise SystemExit(f"{summary}: no {missing} for k_persistent<0, false, ...>")
kb = 1024.0   # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
fetch, write = vals["FETCH_SIZE"][0] * kb / frames, vals["WRITE_SIZE"][0] * kb / frames
rec = {"workload": workload, "kernel": "k_persistent<fused>", "bound": bound,
       "source": f"{kept} (rocprofv3 --pmc passes of tools/profile_gpu.sh, mean per dispatch of k_persistent<0, false, ...>, "
                 f"{frames} frame(s) per dispatch, {vals['SQ_INSTS_VALU'][1]} dispatches)",
       "frames_per_dispatch": frames,
       "per_frame": {"fetch_bytes_reported": fetch, "write_bytes": write, "hbm_bytes": 2.0 * fetch + write,
                     "valu_wave_instructions": vals["SQ_INSTS_VALU"][0] / frames,
                     "tcp_cache_accesses": vals["TCP_TOTAL_CACHE_ACCESSES_sum"][0] / frames,
                     **({"salu_wave_instructions": vals["SQ_INSTS_SALU"][0] / frames} if "SQ_INSTS_SALU" in vals else {}),
                     **({"l2_hit_rate": vals["TCC_HIT_sum"][0] / (vals["TCC_HIT_sum"][0] + vals["TCC_MISS_sum"][0])} if "TCC_HIT_sum" in vals else {})},
       "note": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) loads at 64 B, so the read side is "
               "doubled; WRITE_SIZE is exact"}
if "TA_BUSY_avr" in busy and busy.get("GRBM_GUI_ACTIVE"):
    # TA_BUSY counts, per CU, the cycles its texture-address unit (the front of the L1 path) is busy; GRBM_GUI_ACTIVE is summed over
    # the 8 XCDs.  Both are from one pass, so their quotient is the busy share of that dispatch's cycles.
    rec["l1_path_busy"] = {"ta_busy_avr_cycles": busy["TA_BUSY_avr"], "ta_busy_max_cycles": busy.get("TA_BUSY_max"),
                           "gui_active_cycles_per_xcd": busy["GRBM_GUI_ACTIVE"] / 8.0,
                           "ta_busy_frac": busy["TA_BUSY_avr"] / (busy["GRBM_GUI_ACTIVE"] / 8.0)}
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"counters_{workload}.json")
json.dump(rec, open(out, "w"), indent=1)
print(out, json.dumps(rec["per_frame"]))
