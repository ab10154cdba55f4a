#!/usr/bin/env python3
"""BUILD CONTAINER ONLY: time the UNMODIFIED reference fused shader on Mesa llvmpipe
for the bench workload (baseline B1 of BASELINE.md).  Needs /root/reference."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import bench
from oracles import OracleA, OracleB
name = sys.argv[1] if len(sys.argv) > 1 else "atrium_1080p"
model, camera, W, H, tex = bench.build_workload(name)
sc = model.scene_arrays()
dirs, orgs = camera.primary_rays(W, H)
B = OracleB()
out, med = B.fused(sc, dirs, orgs, model.has_tex_coords(), model.has_normals(), tex[0], tex[1], warm=1, reps=3)
A = OracleA()
t0 = time.time(); a = A.render(sc, dirs, orgs, model.has_tex_coords(), model.has_normals(), tex[0], tex[1], nthreads=os.cpu_count()); ta = time.time() - t0
rays = W * H + int((a["hit_id"] != 0xFFFFFFFF).sum())
err = np.abs(np.clip(out.reshape(-1, 4)[:, :3], 0, 1) - np.clip(a["rgba"][:, :3], 0, 1)).max(1) * 255
print(json.dumps({"workload": name, "renderer": B.renderer, "threads": int(os.environ.get("LP_NUM_THREADS", os.cpu_count())),
                  "llvmpipe_ms_per_frame": med * 1e3, "llvmpipe_mrays_s": rays / med / 1e6,
                  "oracle_a_ms_per_frame": ta * 1e3, "oracle_a_mrays_s": rays / ta / 1e6, "oracle_a_threads": os.cpu_count(),
                  "rays_per_frame": rays, "pixels_over_1_255_vs_oracle_a": int((err > 1).sum()), "max_err_255": float(err.max())}))
