#!/usr/bin/env python3
"""Experiment (GPU box): time of rtamd_assemble_tiles_device (root side of the gather) for a 1080p / 4K RGBA32F frame."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd.renderer import HipRayTracer
dev = torch.device("cuda", 0)
tr = HipRayTracer(0)
for W, H in ((1920, 1080), (3840, 2160)):
    for world in (2, 8):
        per = tr.tiles_per_rank(W, H, world) * 1024
        src = torch.rand((world * per, 4), dtype=torch.float32, device=dev)
        dst = torch.empty((W * H, 4), dtype=torch.float32, device=dev)
        st = torch.cuda.Stream(device=dev); tr.set_stream(st.cuda_stream)
        for _ in range(10): tr.assemble_tiles_device(src.data_ptr(), W, H, world, 4, dst.data_ptr())
        torch.cuda.synchronize(); K = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(K): tr.assemble_tiles_device(src.data_ptr(), W, H, world, 4, dst.data_ptr())
        e1.record(st); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / K * 1e3
        print(f"{W}x{H} world {world}: {us:.1f} us per frame, {2 * W * H * 16 / us / 1e6:.2f} TB/s", flush=True)
