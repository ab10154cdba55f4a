#!/usr/bin/env python3
"""Would binning the rays of config 5's continuation lists pay?  (GPU box)
Takes the ray lists of one sample of the multi-bounce chain (atrium, 3840x2160): shadow, reflect and ambient
directions of bounce 1 and 2, and traces each list (rtamd_trace_device, kernel ms from HIP events, median of 7)
 a) as the chain issues it (pixel order),
 b) stably binned by direction octant (origin order kept inside a bin),
 c) binned by octant, then by the Morton code of the origin's cell in a 64^3 grid over the scene box.
The permutation and the gather are done with torch here and are NOT in the figures: they are the price a
binning pass would add (12 B read for the key + 24 B gathered + 16 B of records scattered back per ray, and a sort)."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import bench  # noqa: E402
from raytracer_amd.multibounce import MultiBounce  # noqa: E402
from raytracer_amd.renderer import HipRayTracer  # noqa: E402


def morton3(c):
    def spread(v):
        v = v & 0x3FF
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)


def main():
    W, H = 3840, 2160
    n = W * H
    model, camera, _, _, tex = bench.build_workload("atrium_1080p")
    camera.aspect = W / H
    sc = model.scene_arrays()
    tr = HipRayTracer(0)
    tr.upload_scene(sc)
    rng = np.random.default_rng(3)
    v = rng.normal(size=(512, 512, 3)); v /= np.linalg.norm(v, axis=-1, keepdims=True)
    tr.upload_noise_texture(np.clip(np.round((v * 0.5 + 0.5) * 255), 0, 255).astype(np.uint8))
    mb = MultiBounce(tr, W, H, camera.inv_proj_view(), camera.position, (0.0, 12.0, 0.0), model.has_normals(), depth=3, lanes=1)
    dev = torch.device("cuda", 0)
    pos = torch.tensor(np.asarray(sc["positions"], np.float32).reshape(-1, 3), device=dev)
    lo, hi = pos.min(0).values, pos.max(0).values
    out = []
    hits = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    hits_b = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    for s in (0, 1):      # even sample: mirror continuation; odd: ambient
        rec = {}
        mb.sample(s, record=rec)
        mb.synchronize()
        tr.set_stream(0)
        torch.cuda.synchronize()
        for depth in (1, 2):
            org = rec[(depth, "org")]
            for kind in ("shadow_dir", "reflect_dir" if s % 2 == 0 else "ambient_dir"):
                d = rec[(depth, kind)]
                live = (d != 0).any(1)
                octant = ((d[:, 0] < 0).to(torch.int64) | ((d[:, 1] < 0).to(torch.int64) << 1) | ((d[:, 2] < 0).to(torch.int64) << 2))
                octant = torch.where(live, octant, torch.full_like(octant, 8))      # NO_RAY entries last
                cell = ((org - lo) / (hi - lo).clamp_min(1e-20) * 64).clamp(0, 63).to(torch.int64)
                key_c = (octant << 30) | morton3(cell)
                perms = {"as issued": None, "by octant": torch.argsort(octant, stable=True), "by octant, then origin cell": torch.argsort(key_c, stable=True)}
                row = {"sample": s, "bounce": depth, "list": kind, "rays": int(live.sum())}
                ref = None
                for name, perm in perms.items():
                    o2, d2 = (org, d) if perm is None else (org[perm].contiguous(), d[perm].contiguous())
                    torch.cuda.synchronize()
                    ms = []
                    for _ in range(9):
                        tr.trace_device(n, o2.data_ptr(), d2.data_ptr(), hits_b.data_ptr())
                        tr.synchronize()
                        ms.append(tr.last_kernel_ms())
                    uniform, util, thr = tr.last_launch_coherence()
                    if perm is None:
                        hits.copy_(hits_b); ref = hits
                    else:       # the records, scattered back, must be the bits of the list as issued
                        back = torch.empty_like(hits_b); back[perm] = hits_b
                        assert torch.equal(back.view(torch.int32), ref.view(torch.int32)), (name, kind)
                    row[name] = {"ms": round(float(np.median(ms[2:])), 4), "uniform_visits": round(uniform, 3), "lane_util": round(util, 3), "dyn_thresh": thr}
                # what the binning itself would cost with library primitives (torch: rocPRIM radix sort of a 32-bit key with its index,
                # gather of the 24-byte rays, scatter of the 16-byte records), HIP events, median of 5
                key32 = key_c.to(torch.int32)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                cost = []
                for _ in range(6):
                    ev[0].record(); perm = torch.sort(key32, stable=True).indices
                    ev[1].record(); o2 = org[perm]; d2 = d[perm]
                    ev[2].record(); back = torch.empty_like(hits_b); back[perm] = hits_b
                    ev[3].record(); torch.cuda.synchronize()
                    cost.append([ev[k].elapsed_time(ev[k + 1]) for k in range(3)])
                c = np.median(np.array(cost[1:]), axis=0)
                row["binning_cost_ms"] = {"sort": round(float(c[0]), 3), "gather": round(float(c[1]), 3), "scatter_back": round(float(c[2]), 3)}
                out.append(row)
                print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
