#!/usr/bin/env python3
"""GPU box: what ONE rank's tile share of the 1080p bench frame costs per frame when the frames of a gather are rendered by
one batched launch (rtamd_render_batch_device), for tile_world = 1, 2, 4, 8, G frames per launch and S alternating streams.
Render + display pass only (no collective): the per-rank render time that bounds the N-GPU frame rate.
Usage: tools/share_cost.py [workload]"""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from raytracer_amd.renderer import TILE_PIXELS, HipRayTracer

wl = sys.argv[1] if len(sys.argv) > 1 else "atrium_1080p"
model, camera, W, H, tex = bench.build_workload(wl)
sc = model.scene_arrays()
dev = torch.device("cuda", 0)
tr = HipRayTracer(0)
tr.upload_scene(sc)
for slot, t in enumerate(tex):
    if t is not None:
        tr.upload_texture(slot, t)
n = W * H
d_dir = torch.empty((n, 3), dtype=torch.float32, device=dev)
tr.generate_rays_device(W, H, camera.inv_proj_view(), d_dir.data_ptr()); tr.synchronize()
d_org = torch.tensor(np.asarray(camera.position, np.float32), device=dev).repeat(n, 1).contiguous()
has_tc, has_n = model.has_tex_coords(), model.has_normals()
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
res = []
for world in (1, 2, 4, 8):
    n_out = n if world == 1 else tr.tiles_per_rank(W, H, world) * TILE_PIXELS
    for G in (1, 4, 8):
        for S in (1, 2, 3):
            bufs = [torch.zeros((G * n_out, 4), dtype=torch.float32, device=dev) for _ in range(S)]
            out8 = [torch.zeros((G * n_out,), dtype=torch.int32, device=dev) for _ in range(S)]
            def batch(k):
                f = k % S
                tr.set_stream(streams[f].cuda_stream)
                tr.render_batch_device(W, H, [{"dir_ptr": d_dir.data_ptr(), "org_ptr": d_org.data_ptr(), "rgba_ptr": bufs[f].data_ptr() + g * n_out * 16} for g in range(G)],
                                       has_tex_coords=has_tc, has_normals=has_n, tile_rank=0, tile_world=world)
                tr.display_rgba8_device(bufs[f].data_ptr(), G * n_out, out8[f].data_ptr())
            for k in range(6):
                batch(k)
            torch.cuda.synchronize()
            nb = max(12, 240 // G)
            t0 = time.perf_counter()
            for k in range(nb):
                batch(k)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / (nb * G) * 1e6
            res.append({"tile_world": world, "frames_per_launch": G, "streams": S, "us_per_frame": round(us, 1)})
            print(json.dumps(res[-1]), flush=True)
tr.set_stream(0)
