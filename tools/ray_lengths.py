#!/usr/bin/env python3
"""How long is the longest ray of a frame, and of its slowest 8x8 blocks?  (GPU box.)
The per-pixel test counters of the fused kernel (2 box tests per inner visit, primary + shadow ray of the pixel) give each
pixel's own serial chain; a wave's iteration count (tools/stamps.py) can only be shortened down to that by any re-distribution
of its rays.  Usage: tools/ray_lengths.py [--workload W]"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="atrium_1080p")
    args = ap.parse_args()
    import torch
    import bench
    from raytracer_amd.renderer import HipRayTracer
    model, camera, W, H, tex = bench.build_workload(args.workload)
    dev = torch.device("cuda", 0)
    tr = HipRayTracer(0)
    tr.upload_scene(model.scene_arrays())
    for slot, t in enumerate(tex):
        if t is not None:
            tr.upload_texture(slot, t)
    n = W * H
    d_counts = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    d_t = torch.zeros((n,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    cam = (camera.inv_proj_view(), camera.position)
    tr.render_device(W, H, camera=cam, has_tex_coords=model.has_tex_coords(), has_normals=model.has_normals(),
                     counts_ptr=d_counts.data_ptr(), hit_t_ptr=d_t.data_ptr())
    tr.synchronize()
    c = d_counts.cpu().numpy().astype(np.int64).reshape(H, W, 2)
    visits = c[..., 0] // 2          # inner visits (each tests two boxes; the root test counts one box per ray)
    tris = c[..., 1]
    out = {"workload": args.workload,
           "inner_visits_per_pixel": {"mean": float(visits.mean()), "p50": float(np.percentile(visits, 50)), "p99": float(np.percentile(visits, 99)),
                                      "p99.9": float(np.percentile(visits, 99.9)), "max": int(visits.max())},
           "tri_tests_per_pixel": {"mean": float(tris.mean()), "p99": float(np.percentile(tris, 99)), "max": int(tris.max())}}
    # per 8x8 block: the longest pixel and the mean
    hb, wb = H // 8, W // 8
    v = visits[: hb * 8, : wb * 8].reshape(hb, 8, wb, 8)
    bmax = v.max(axis=(1, 3)); bmean = v.mean(axis=(1, 3))
    order = np.argsort(bmax.ravel())[::-1][:12]
    out["blocks_by_longest_pixel"] = [{"px": [int(i % wb) * 8, int(i // wb) * 8], "max_visits": int(bmax.ravel()[i]), "mean_visits": round(float(bmean.ravel()[i]), 1),
                                       "pixels_over_half_max": int((v[i // wb, :, i % wb, :] > bmax.ravel()[i] / 2).sum())} for i in order]
    for px, py in ((904, 256), (784, 272)):
        blk = visits[py:py + 8, px:px + 8]
        out[f"block_{px}_{py}"] = {"max": int(blk.max()), "mean": round(float(blk.mean()), 1), "sorted_top16": sorted(blk.ravel().tolist())[-16:]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
