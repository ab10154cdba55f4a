#!/usr/bin/env python3
"""BASELINE config 5 on the GPU box (throughput only — SURVEY section 8d: no reference colour exists for it):
Sponza-class atrium, 3840x2160, S samples per pixel, depth <= 3.  Per sample: jittered primaries
(rtamd_generate_rays_offset_device, R2 sequence) -> rtamd_trace_device; at every hit the dispatcher
(ray_dispatcher.frag) emits a shadow ray to a point light and one continuation — the mirror direction for even
sample indices, the ambient direction normalize(n + noise*2-1) for odd ones — and rtamd_surface_from_hits_device
turns the continuation's hit records into the next G-buffer.  Every stage is parity-tested against the oracle
(tests/test_gpu_parity.py).  Samples rotate over --lanes streams with their own buffers (frames in flight).
A ray = one traversal that was actually started (NO_RAY entries are not counted).
Usage: tools/bench_c5.py [--spp 8] [--width 3840 --height 2160] [--depth 3]"""
import argparse, json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from raytracer_amd.renderer import HipRayTracer

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--lanes", type=int, default=4, help="samples in flight (streams)")
args = ap.parse_args()
W, H, n = args.width, args.height, args.width * args.height
dev = torch.device("cuda", 0)
model, camera, _, _, tex = bench.build_workload("atrium_1080p")
camera.aspect = W / H
sc = model.scene_arrays()
has_n = model.has_normals()
tr = HipRayTracer(0)
tr.upload_scene(sc)
rng = np.random.default_rng(3)
v = rng.normal(size=(512, 512, 3)); v /= np.linalg.norm(v, axis=-1, keepdims=True)
tr.upload_noise_texture(np.clip(np.round((v * 0.5 + 0.5) * 255), 0, 255).astype(np.uint8))
inv = camera.inv_proj_view()
cam = tuple(float(x) for x in camera.position)
light = (0.0, 12.0, 0.0)


class Lane:   # one sample in flight: a stream and its buffers
    def __init__(self):
        self.stream = torch.cuda.Stream(device=dev)
        f = lambda c: torch.empty((n, c), dtype=torch.float32, device=dev)
        self.dir, self.hits, self.pos, self.nm = f(3), f(4), f(3), f(4)
        self.org2, self.shadow, self.reflect, self.ambient, self.shadow_hits = f(3), f(3), f(3), f(3), f(4)
        self.cam_org = torch.tensor(cam, dtype=torch.float32, device=dev).repeat(n, 1).contiguous()
        self.rays = torch.zeros((), dtype=torch.int64, device=dev)


lanes = [Lane() for _ in range(max(1, min(8, args.lanes)))]
G = 0.7548776662466927   # R2 sequence (plastic number)


def sample(s, lane):
    with torch.cuda.stream(lane.stream):
        tr.set_stream(lane.stream.cuda_stream)
        ox, oy = (0.5 + s * G) % 1.0, (0.5 + s * G * G) % 1.0
        tr.generate_rays_offset_device(W, H, inv, ox, oy, lane.dir.data_ptr())
        tr.trace_device(n, lane.cam_org.data_ptr(), lane.dir.data_ptr(), lane.hits.data_ptr())
        lane.rays += n
        org, d = lane.cam_org, lane.dir
        for depth in range(1, args.depth + 1):
            tr.surface_from_hits_device(n, org.data_ptr(), d.data_ptr(), lane.hits.data_ptr(), has_n, lane.pos.data_ptr(), lane.nm.data_ptr())
            nos = ((7 * s + 3 * depth) % 512 / 512, (11 * s + 5 * depth) % 512 / 512, W / 512, H / 512)
            tr.dispatch_rays_device(W, H, lane.pos.data_ptr(), lane.nm.data_ptr(), light, cam, nos, org_ptr=lane.org2.data_ptr(),
                                    shadow_dir_ptr=lane.shadow.data_ptr(), reflect_dir_ptr=lane.reflect.data_ptr(), ambient_dir_ptr=lane.ambient.data_ptr())
            surfaces = (lane.nm[:, 3] != 1e30).sum()
            tr.trace_device(n, lane.org2.data_ptr(), lane.shadow.data_ptr(), lane.shadow_hits.data_ptr())
            lane.rays += surfaces
            if depth < args.depth:
                cont = lane.reflect if s % 2 == 0 else lane.ambient
                # the continuation's origin / direction must survive the next dispatch: swap buffers
                lane.dir2 = getattr(lane, "dir2", None)
                if lane.dir2 is None:
                    lane.dir2 = torch.empty_like(lane.dir); lane.org3 = torch.empty_like(lane.dir)
                lane.dir2.copy_(cont); lane.org3.copy_(lane.org2)
                tr.trace_device(n, lane.org3.data_ptr(), lane.dir2.data_ptr(), lane.hits.data_ptr())
                lane.rays += surfaces
                org, d = lane.org3, lane.dir2


for s in range(len(lanes)):
    sample(s, lanes[s % len(lanes)])
torch.cuda.synchronize()
for l in lanes:
    l.rays.zero_()
t0 = time.perf_counter()
for s in range(args.spp):
    sample(s, lanes[s % len(lanes)])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rays = int(sum(int(l.rays.item()) for l in lanes))
print(json.dumps({"metric": "Mrays/s", "value": round(rays / dt / 1e6, 1), "unit": "Mrays/s", "n_gpus": 1,
                  "config": {"workload": f"C5: atrium {len(sc['triangles'])} triangles, {W}x{H}, {args.spp} spp, depth <= {args.depth}, "
                                         "primary + (shadow + mirror|ambient continuation) per bounce", "rays": rays,
                             "rays_per_pixel_sample": round(rays / (n * args.spp), 3), "samples_in_flight": len(lanes)},
                  "seconds": round(dt, 4), "ms_per_sample_frame": round(dt / args.spp * 1e3, 3), "data": "synthetic", "dtype": "f32"}))
