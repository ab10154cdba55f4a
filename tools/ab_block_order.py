#!/usr/bin/env python3
"""Config 5's ray lists are in scan-line pixel order: a work item of 64 consecutive rays is a 64 x 1 strip of the image.
What would 8 x 8 blocks (the fused kernel's item shape) give?  (GPU box.)  Each list of one sample of the chain (atrium,
3840x2160) is traced as issued and permuted into 8x8-block order (torch gather, not in the figures); records compared bit for bit."""
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
idx = torch.arange(n, device=dev).view(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1)     # 8x8 blocks, row-major inside
hits_a = torch.zeros((n, 4), dtype=torch.float32, device=dev); hits_b = torch.zeros_like(hits_a)
def timed(o, d, out, image=False):
    ms = []
    for _ in range(9):
        if image:
            tr.trace_image_batch_device([(n, o.data_ptr(), d.data_ptr(), out.data_ptr())], W)
        else:
            tr.trace_device(n, o.data_ptr(), d.data_ptr(), out.data_ptr())
        tr.synchronize(); ms.append(tr.last_kernel_ms())
    return round(float(np.median(ms[2:])), 4), tr.last_launch_coherence()
tot = [0.0, 0.0, 0.0]
for s in (0, 1):
    rec = {}
    mb.sample(s, record=rec); mb.synchronize(); tr.set_stream(0); torch.cuda.synchronize()
    lists = [("primary", mb.lanes[0].cam_org, rec[(0, "dir")])]
    for depth in (1, 2, 3):
        lists.append((f"bounce {depth} shadow", rec[(depth, "org")], rec[(depth, "shadow_dir")]))
        if depth < 3:
            lists.append((f"bounce {depth} " + ("reflect" if s % 2 == 0 else "ambient"), rec[(depth, "org")], rec[(depth, "reflect_dir" if s % 2 == 0 else "ambient_dir")]))
    for name, o, d in lists:
        a, ca = timed(o, d, hits_a)
        o2, d2 = o[idx].contiguous(), d[idx].contiguous(); torch.cuda.synchronize()
        b, cb = timed(o2, d2, hits_b)
        back = torch.empty_like(hits_b); back[idx] = hits_b
        assert torch.equal(back.view(torch.int32), hits_a.view(torch.int32)), name
        hits_b.zero_(); torch.cuda.synchronize()
        c, cc = timed(o, d, hits_b, image=True)       # the library's own block items: rtamd_trace_image_batch_device on the list as issued
        assert torch.equal(hits_b.view(torch.int32), hits_a.view(torch.int32)), name
        tot[0] += a; tot[1] += b; tot[2] += c
        print(json.dumps({"sample": s, "list": name, "scan-line ms": a, "8x8 blocks ms": b, "rtamd_trace_image_batch_device ms": c, "uniform visits": [round(ca[0], 3), round(cb[0], 3)], "lane util": [round(ca[1], 3), round(cb[1], 3)], "dyn_thresh": [ca[2], cb[2]]}), flush=True)
print(json.dumps({"all lists of two samples": {"scan-line ms": round(tot[0], 3), "8x8 blocks ms": round(tot[1], 3), "rtamd_trace_image_batch_device ms": round(tot[2], 3)}}))
