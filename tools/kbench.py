#!/usr/bin/env python3
"""Kernel A/B bench (GPU box): one workload, many (library build, option) arms,
interleaved rounds in one process per library, kernel time from HIP events.
Usage: tools/kbench.py [--workload W] [--rounds N] arm [arm ...]
  arm = name[:key=value[,key=value...]]   keys: variant, dyn_thresh, num_queues, leaf_thresh, queue_stride, queue_block,
        blocks (max_blocks_per_cu), anyhit, mode(fused|trace|incoherent)
Libraries: RTAMD_DEVICE_LIB env (one process = one library)."""
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
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--inflight", type=int, default=1, help=">1: frames alternate over that many streams and the figure is wall ms per frame over 24 frames")
    ap.add_argument("arms", nargs="+")
    args = ap.parse_args()
    import torch
    import bench
    from raytracer_amd.renderer import HipRayTracer
    model, camera, W, H, tex = bench.build_workload(args.workload)
    sc = model.scene_arrays()
    dev = torch.device("cuda", 0)
    tr = HipRayTracer(0)
    tr.upload_scene(sc)
    for slot, t in enumerate(tex):
        if t is not None:
            tr.upload_texture(slot, t)
    n = W * H
    d_dir = torch.empty((n, 3), dtype=torch.float32, device=dev)
    tr.generate_rays_device(W, H, camera.inv_proj_view(), d_dir.data_ptr())
    tr.synchronize()
    d_org = torch.tensor(np.asarray(camera.position, np.float32), device=dev).repeat(n, 1).contiguous()
    d_rgba = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    d_hits = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    # incoherent rays: shuffled primaries (same ray set, no spatial coherence inside a wave)
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    d_dir_sh = d_dir[perm].contiguous()
    torch.cuda.synchronize()
    arms = []
    for a in args.arms:
        name, _, kv = a.partition(":")
        opts = dict(x.split("=") for x in kv.split(",")) if kv else {}
        arms.append((name, opts))
    res = {name: [] for name, _ in arms}
    coh = {}
    ref_out = {}
    import time
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.inflight))]
    outs = [torch.zeros((n, 4), dtype=torch.float32, device=dev) for _ in streams]
    has_tc, has_n = model.has_tex_coords(), model.has_normals()
    for r in range(args.rounds + 2):
        for name, o in arms:
            tr.set_option("kernel_variant", int(o.get("variant", 0)))
            mode = o.get("mode", "fused")
            tr.set_option("dyn_thresh", int(o.get("dyn_thresh", 0)))
            tr.set_option("num_queues", int(o.get("num_queues", 8)))
            tr.set_option("leaf_thresh", int(o.get("leaf_thresh", 0)))
            tr.set_option("queue_stride", int(o.get("queue_stride", 64)))
            tr.set_option("max_blocks_per_cu", int(o.get("blocks", 0)))
            tr.set_option("queue_block", int(o.get("queue_block", 256 if o.get("mode", "fused") == "fused" else 1)))
            tr.set_option("shadow_anyhit", int(o.get("anyhit", 1)))
            mode = o.get("mode", "fused")
            if args.inflight > 1:
                for rep in range(2):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for k in range(24):
                        tr.set_stream(streams[k % len(streams)].cuda_stream)
                        if mode == "fused":
                            tr.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=has_tc, has_normals=has_n, rgba_ptr=outs[k % len(streams)].data_ptr())
                        else:
                            tr.trace_device(n, d_org.data_ptr(), (d_dir if mode == "trace" else d_dir_sh).data_ptr(), outs[k % len(streams)].data_ptr())
                    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 24 * 1e3
                if r >= 2:
                    res[name].append(dt)
                coh[name] = tr.last_launch_coherence() if int(o.get("variant", 0)) == 0 else (-1.0, -1.0, 0)
                continue
            for _ in range(3):   # the schedule of a launch comes from the previous launch of the same arm
                if mode == "fused":
                    tr.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=has_tc, has_normals=has_n, rgba_ptr=d_rgba.data_ptr())
                elif mode == "trace":
                    tr.trace_device(n, d_org.data_ptr(), d_dir.data_ptr(), d_hits.data_ptr())
                else:
                    tr.trace_device(n, d_org.data_ptr(), d_dir_sh.data_ptr(), d_hits.data_ptr())
            ms = tr.last_kernel_ms()
            # every arm must produce the bits of the first arm of its mode (results never depend on scheduling)
            out_now = (d_rgba if mode == "fused" else d_hits).view(torch.int32)
            if mode not in ref_out:
                ref_out[mode] = out_now.clone()
            elif not torch.equal(ref_out[mode], out_now):
                raise SystemExit(f"arm {name}: output differs from the first {mode} arm's ({int((ref_out[mode] != out_now).any(1).sum())} pixels)")
            if r >= 2:
                res[name].append(ms)
            coh[name] = tr.last_launch_coherence() if int(o.get("variant", 0)) == 0 else (-1.0, -1.0, 0)
    lib = os.environ.get("RTAMD_DEVICE_LIB", "default")
    for name, _ in arms:
        v = np.array(res[name])
        print(json.dumps({"lib": os.path.basename(lib), "arm": name, "median_ms": round(float(np.median(v)), 4), "min_ms": round(float(v.min()), 4), "uniform": round(coh[name][0], 3), "lane_util": round(coh[name][1], 3), "dyn_thresh": coh[name][2]}))


if __name__ == "__main__":
    main()
