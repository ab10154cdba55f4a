#!/usr/bin/env python3
"""Experiment (GPU box): frames in flight.  Wall time per frame when K frames are issued
round-robin on S streams of ONE context (per-stream launch scratch inside the library),
each stream with its own output buffer."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch, bench
from raytracer_amd.renderer import HipRayTracer
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium_1080p"
model, camera, W, H, tex = bench.build_workload(wl)
sc = model.scene_arrays(); dev = torch.device("cuda", 0)
n = W * H
tr = HipRayTracer(0); tr.upload_scene(sc)
for slot, t in enumerate(tex):
    if t is not None: tr.upload_texture(slot, t)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
bufs = [torch.zeros((n, 4), dtype=torch.float32, device=dev) for _ in range(4)]
d_dir = torch.empty((n, 3), dtype=torch.float32, device=dev)
tr.generate_rays_device(W, H, camera.inv_proj_view(), d_dir.data_ptr()); tr.synchronize()
d_org = torch.tensor(np.asarray(camera.position, np.float32), device=dev).repeat(n, 1).contiguous()
torch.cuda.synchronize()
K = 300
has_tc, has_n = model.has_tex_coords(), model.has_normals()
for S, EV in ((1, 0), (2, 0), (2, 1), (2, 2), (2, 0), (2, 1), (2, 2)):
    for rep in range(2):
        evs = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            tr.set_stream(streams[k % S].cuda_stream)
            if EV:
                e0 = torch.cuda.Event(enable_timing=(EV == 1)); e1 = torch.cuda.Event(enable_timing=(EV == 1)); e0.record(streams[k % S])
            tr.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=has_tc, has_normals=has_n, rgba_ptr=bufs[k % S].data_ptr())
            if EV:
                e1.record(streams[k % S]); evs.append((e0, e1))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"frames in flight {S}, torch events {EV}: {dt / K * 1e3:.4f} ms per frame, last kernel {tr.last_kernel_ms():.4f} ms", flush=True)
for b in bufs[1:]:
    assert torch.equal(bufs[0], b)
