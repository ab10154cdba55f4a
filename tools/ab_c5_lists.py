#!/usr/bin/env python3
"""Kernel time of ONE launch of config 5's chain (GPU box): the 16 ray lists of a bounce of 8 samples (8 shadow lists + 4 mirror
+ 4 ambient continuations, 3840x2160 each), through rtamd_trace_batch_device (64 consecutive rays per item) and through
rtamd_trace_image_batch_device (8x8-pixel items), HIP-event kernel ms, median of 7.  RTAMD_DEVICE_LIB selects the build."""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch, bench
from raytracer_amd.multibounce import MultiBounce
from raytracer_amd.renderer import HipRayTracer
W, H = 3840, 2160
n = W * H
model, camera, _, _, tex = bench.build_workload("atrium_1080p")
camera.aspect = W / H
tr = HipRayTracer(0); tr.upload_scene(model.scene_arrays())
rng = np.random.default_rng(3)
v = rng.normal(size=(512, 512, 3)); v /= np.linalg.norm(v, axis=-1, keepdims=True)
tr.upload_noise_texture(np.clip(np.round((v * 0.5 + 0.5) * 255), 0, 255).astype(np.uint8))
mb = MultiBounce(tr, W, H, camera.inv_proj_view(), camera.position, (0.0, 12.0, 0.0), model.has_normals(), depth=3, lanes=1)
dev = torch.device("cuda", 0)
recs = []
for s in range(8):
    rec = {}
    mb.sample(s, record=rec); mb.synchronize()
    recs.append({k: v for k, v in rec.items() if k[1] in ("org", "shadow_dir", "reflect_dir", "ambient_dir", "dir")})
tr.set_stream(0); torch.cuda.synchronize()
outs = [torch.zeros((n, 4), dtype=torch.float32, device=dev) for _ in range(16)]
cam_org = mb.lanes[0].cam_org
has_image = hasattr(tr._lib, "rtamd_trace_image_batch_device")
for depth in (0, 1, 2, 3):
    if depth == 0:
        lists = [(cam_org, r[(0, "dir")]) for r in recs]
    else:
        lists = [(r[(depth, "org")], r[(depth, "shadow_dir")]) for r in recs]
        if depth < 3:
            lists += [(r[(depth, "org")], r[(depth, "reflect_dir" if s % 2 == 0 else "ambient_dir")]) for s, r in enumerate(recs)]
    jobs = [(n, o.data_ptr(), d.data_ptr(), outs[k].data_ptr()) for k, (o, d) in enumerate(lists)]
    row = {"lib": os.path.basename(os.environ.get("RTAMD_DEVICE_LIB", "default")), "launch": "primaries" if depth == 0 else f"bounce {depth}", "lists": len(jobs)}
    ref = None
    for name in ("plain", "image"):
        if name == "image" and not has_image:
            continue
        ms = []
        for _ in range(9):
            tr.trace_batch_device(jobs) if name == "plain" else tr.trace_image_batch_device(jobs, W)
            tr.synchronize(); ms.append(tr.last_kernel_ms())
        row[name + " ms"] = round(float(np.median(ms[2:])), 4)
        cur = torch.stack([o.view(torch.int32) for o in outs[:len(jobs)]])
        if ref is None: ref = cur.clone()
        else: assert torch.equal(ref, cur)
    print(json.dumps(row), flush=True)
