#!/usr/bin/env python3
"""BASELINE config 5 on the GPU box (throughput only — SURVEY section 8d: no reference colour exists for it):
Sponza-class atrium, 3840x2160, S samples per pixel, depth <= 3, through raytracer_amd/multibounce.py (the chain
tests/test_gpu_multibounce.py checks buffer by buffer against the oracle).  Samples rotate over --lanes streams
with their own buffers.  A ray = one traversal that was actually started (NO_RAY entries are not counted).
Usage: tools/bench_c5.py [--spp 8] [--width 3840 --height 2160] [--depth 3]"""
import argparse, json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from raytracer_amd.multibounce import MultiBounce
from raytracer_amd.renderer import HipRayTracer

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--lanes", type=int, default=4, help="samples in flight (streams); measured 1 / 2 / 3 / 4 / 6: 6.21 / 6.71 / 6.44 / 6.92 / 6.22 Grays/s (4 = one per default hardware queue)")
args = ap.parse_args()
W, H, n = args.width, args.height, args.width * args.height
model, camera, _, _, tex = bench.build_workload("atrium_1080p")
camera.aspect = W / H
sc = model.scene_arrays()
tr = HipRayTracer(0)
tr.upload_scene(sc)
rng = np.random.default_rng(3)
v = rng.normal(size=(512, 512, 3)); v /= np.linalg.norm(v, axis=-1, keepdims=True)
tr.upload_noise_texture(np.clip(np.round((v * 0.5 + 0.5) * 255), 0, 255).astype(np.uint8))
mb = MultiBounce(tr, W, H, camera.inv_proj_view(), camera.position, (0.0, 12.0, 0.0), model.has_normals(), depth=args.depth, lanes=args.lanes)
for s in range(len(mb.lanes)):
    mb.sample(s)
mb.rays_traced(reset=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(args.spp):
    mb.sample(s)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rays = mb.rays_traced()
print(json.dumps({"metric": "Mrays/s", "value": round(rays / dt / 1e6, 1), "unit": "Mrays/s", "n_gpus": 1,
                  "config": {"workload": f"C5: atrium {len(sc['triangles'])} triangles, {W}x{H}, {args.spp} spp, depth <= {args.depth}, "
                                         "primary + (shadow + mirror|ambient continuation) per bounce", "rays": rays,
                             "rays_per_pixel_sample": round(rays / (n * args.spp), 3), "samples_in_flight": len(mb.lanes)},
                  "seconds": round(dt, 4), "ms_per_sample_frame": round(dt / args.spp * 1e3, 3), "data": "synthetic", "dtype": "f32"}))
