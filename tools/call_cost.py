#!/usr/bin/env python3
"""GPU box: host cost of one C-ABI call (issue rate of tiny launches) with and without the device-buffer range checks."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from common import load_golden
from raytracer_amd.renderer import HipRayTracer
g = load_golden("cube")
tr = HipRayTracer(0)
tr.upload_scene(g["scene"])
dev = torch.device("cuda", 0)
n = 64
org = torch.zeros((n, 3), dtype=torch.float32, device=dev); d = torch.zeros((n, 3), dtype=torch.float32, device=dev); d[:, 2] = -1
hits = torch.zeros((n, 4), dtype=torch.float32, device=dev)
rgba = torch.zeros((n, 4), dtype=torch.float32, device=dev)
for v in (1, 0, 1, 0):
    tr.set_option("validate_buffers", v)
    for name, fn in (("trace_device", lambda: tr.trace_device(n, org.data_ptr(), d.data_ptr(), hits.data_ptr())),
                     ("render_device", lambda: tr.render_device(8, 8, d.data_ptr(), org.data_ptr(), rgba_ptr=rgba.data_ptr())),
                     ("display_rgba8_device", lambda: tr.display_rgba8_device(rgba.data_ptr(), n, hits.data_ptr()))):
        for _ in range(200): fn()
        tr.synchronize(); t0 = time.perf_counter()
        for _ in range(3000): fn()
        t1 = time.perf_counter(); tr.synchronize()
        print(f"validate_buffers={v} {name}: {(t1 - t0) / 3000 * 1e6:.2f} us per call (host issue)", flush=True)
