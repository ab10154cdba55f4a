#!/usr/bin/env python3
"""BUILD CONTAINER ONLY: baseline B1 of BASELINE.md — the UNMODIFIED reference fused shader
(res/shaders/ray_trace_and_shade.frag:195-235, read from /root/reference at run time) timed on Mesa llvmpipe for
bench workloads, on this container's host cores.  The shader text cannot travel to the GPU box, so the record is
made here and committed: profiles/llvmpipe_baseline.json, one entry per workload, which bench.py copies into
cpu_baseline.reference_llvmpipe.  Usage: tools/llvmpipe_baseline.py [workload ...]   (default: f16_640x480 atrium_1080p)"""
import json, os, platform, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import bench
from oracles import OracleA, OracleB

OUT = os.path.join(REPO, "profiles", "llvmpipe_baseline.json")
names = sys.argv[1:] or ["f16_640x480", "atrium_1080p"]
record = json.load(open(OUT)) if os.path.exists(OUT) else {}
B = OracleB()
A = OracleA()
cores = os.cpu_count()
threads = int(os.environ.get("LP_NUM_THREADS", min(cores, 16)))   # Mesa 23.2 caps llvmpipe at 16 threads
for name in names:
    model, camera, W, H, tex = bench.build_workload(name)
    model.build_bvh(cores)
    sc = model.scene_arrays()
    dirs, orgs = camera.primary_rays(W, H)
    has_tc, has_n = model.has_tex_coords(), model.has_normals()
    out, med = B.fused(sc, dirs, orgs, has_tc, has_n, tex[0], tex[1], warm=2, reps=5)   # glFinish-bracketed draws: median of 5 after 2 warm-ups
    t0 = time.time(); a = A.render(sc, dirs, orgs, has_tc, has_n, tex[0], tex[1], nthreads=cores); ta = time.time() - t0
    rays = W * H + int((a["hit_id"] != 0xFFFFFFFF).sum())
    record[name] = {"workload": name, "shader": "res/shaders/ray_trace_and_shade.frag (unmodified, read from the reference tree at run time)",
                    "renderer": B.renderer, "where": "build container (the reference does not travel to the GPU box)",
                    "host_cores": cores, "llvmpipe_threads": threads, "cpu": platform.processor() or platform.machine(),
                    "frame": f"{W}x{H}", "rays_per_frame": rays, "ms_per_frame": round(med * 1e3, 2), "mrays_per_s": round(rays / med / 1e6, 3),
                    "timing": "median of 5 glFinish-bracketed draws after 2 warm-ups",
                    "oracle_a_same_cores": {"ms_per_frame": round(ta * 1e3, 2), "mrays_per_s": round(rays / ta / 1e6, 3), "threads": cores}}
    print(name, json.dumps(record[name]))
    json.dump(record, open(OUT, "w"), indent=1, sort_keys=True)
