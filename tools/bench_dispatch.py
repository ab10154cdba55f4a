#!/usr/bin/env python3
"""Secondary-ray dispatcher (k_dispatch_rays, ray_dispatcher.frag) on the GPU box: GB/s against the
HBM roofline.  Algorithmic bytes per pixel: 12 (position) + 16 (normal/material) in, 4 x 12 out = 76 B
(noise texels are a 48 KB cache-resident table).  Prints one JSON line per frame size."""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from raytracer_amd.renderer import HipRayTracer

dev = torch.device("cuda", 0)
tr = HipRayTracer(0)
rng = np.random.default_rng(3)
v = rng.normal(size=(512, 512, 3)); v /= np.linalg.norm(v, axis=-1, keepdims=True)
tr.upload_noise_texture(np.clip(np.round((v * 0.5 + 0.5) * 255), 0, 255).astype(np.uint8))   # blue_noise.png is 512 x 512
st = torch.cuda.Stream(device=dev); tr.set_stream(st.cuda_stream)
for w, h in ((1920, 1080), (3840, 2160), (7680, 4320)):
    n = w * h
    g = torch.Generator(device=dev).manual_seed(1)
    pos = (torch.rand((n, 3), device=dev, generator=g) * 40 - 20).contiguous()
    nm = torch.randn((n, 4), device=dev, generator=g)
    nm[:, :3] /= nm[:, :3].norm(dim=1, keepdim=True)
    nm[:, 3] = torch.where(torch.rand(n, device=dev, generator=g) < 0.1, torch.tensor(1e30, device=dev), torch.zeros((), device=dev))
    outs = [torch.empty((n, 3), dtype=torch.float32, device=dev) for _ in range(4)]
    args = (w, h, pos.data_ptr(), nm.data_ptr(), (3.0, 20.0, -4.0), (0.0, 1.0, 5.0), (5 / 512, 9 / 512, w / 512, h / 512))
    kw = dict(org_ptr=outs[0].data_ptr(), shadow_dir_ptr=outs[1].data_ptr(), reflect_dir_ptr=outs[2].data_ptr(), ambient_dir_ptr=outs[3].data_ptr())
    for _ in range(5):
        tr.dispatch_rays_device(*args, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 50
    e0.record(st)
    for _ in range(K):
        tr.dispatch_rays_device(*args, **kw)
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    gbs = 76 * n / (ms * 1e-3) / 1e9
    print(json.dumps({"kernel": "k_dispatch_rays", "frame": f"{w}x{h}", "ms": round(ms, 4), "Mpixels_per_s": round(n / ms / 1e3, 1),
                      "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4)}}), flush=True)
