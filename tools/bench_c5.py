#!/usr/bin/env python3
"""BASELINE config 5 on the GPU box (throughput only — SURVEY section 8d: no reference colour exists for it):
Sponza-class atrium, 3840x2160, S samples per pixel, depth <= 3, through raytracer_amd/multibounce.py (the chain
tests/test_gpu_multibounce.py checks buffer by buffer against the oracle).  Samples rotate over --lanes streams
with their own buffers.  A ray = one traversal that was actually started (NO_RAY entries are not counted).
Several GPUs (`python -m torch.distributed.run --nproc-per-node N tools/bench_c5.py --gpus N`): every rank traces its
interleaved 32x32 tiles of the frame (scene replicated, all samples of a pixel on the GPU that owns its tile, no
collective in the chain: there is no frame to assemble, the records ARE the output); barrier + synchronise on both sides of
the timed region, time = MAX over ranks, rays = SUM.  RTAMD_BENCH_REHEARSE=1 puts every rank on cuda:0 over gloo (a code-path
rehearsal on a one-GPU box, not a measurement).
Usage: tools/bench_c5.py [--spp 8] [--width 3840 --height 2160] [--depth 3] [--gpus N]"""
import argparse, json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from raytracer_amd.multibounce import MultiBounce
from raytracer_amd.renderer import HipRayTracer

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--batch", type=int, default=-1, help="samples in lock step per call of the chain's C entry point (rtamd_multibounce_sample_batch_device, "
                                                      "1..8: every stage's traversals of those samples are ONE rtamd_trace_image_batch_device launch); 0: the chain issued "
                                                      "from Python, one sample per lane, a launch per traversal (MultiBounce.sample); -1 = auto: 1 for the whole frame on one GPU "
                                                      "(round 3, lists traced as images with the waves' own re-fill choice: 1 / 2 / 4 / 8 samples per call on 4 / 2 / 2 / 2 streams "
                                                      "measure 8.17 / 7.85 / 7.69 / 7.59 Grays/s), 8 for a tile share (1/8 share: 1.10 -> 0.87 ms per sample frame)")
ap.add_argument("--groups", type=int, default=0, help="with --batch: batches in flight (streams); 0 = auto: 4 with one sample per call (one per default hardware queue), else 2")
ap.add_argument("--share-of", type=int, default=0, help="one process, one GPU: trace only rank 0's tile share of a split over that many ranks "
                                                         "(what one rank of an N-GPU run has to do; for scaling expectations)")
ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="rtamd_set_option before anything is launched (A/B runs)")
ap.add_argument("--lanes", type=int, default=4, help="samples in flight (streams); measured 1 / 2 / 3 / 4 / 6: 6.21 / 6.71 / 6.44 / 6.92 / 6.22 Grays/s (4 = one per default hardware queue)")
args = ap.parse_args()
W, H, n = args.width, args.height, args.width * args.height
world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
if world != args.gpus and world > 1:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
rehearse = world > 1 and os.environ.get("RTAMD_BENCH_REHEARSE") == "1"
if rehearse:
    local_rank = 0
torch.cuda.set_device(local_rank)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo") if rehearse else dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
if args.share_of > 1 and world == 1:
    world_for_tiles, rank_for_tiles = args.share_of, 0
else:
    world_for_tiles, rank_for_tiles = world, rank
if args.batch < 0:
    args.batch = 1 if world_for_tiles == 1 else 8
if args.groups <= 0:
    args.groups = 4 if args.batch == 1 else 2
model, camera, _, _, tex = bench.build_workload("atrium_1080p")
camera.aspect = W / H
sc = model.scene_arrays()
tr = HipRayTracer(local_rank)
for kv in args.option:
    tr.set_option(kv.partition('=')[0], int(kv.partition('=')[2]))
tr.upload_scene(sc)
rng = np.random.default_rng(3)
v = rng.normal(size=(512, 512, 3)); v /= np.linalg.norm(v, axis=-1, keepdims=True)
tr.upload_noise_texture(np.clip(np.round((v * 0.5 + 0.5) * 255), 0, 255).astype(np.uint8))
mb = MultiBounce(tr, W, H, camera.inv_proj_view(), camera.position, (0.0, 12.0, 0.0), model.has_normals(), depth=args.depth, lanes=(args.batch * args.groups if args.batch else args.lanes),
                 device=local_rank, tile_rank=rank_for_tiles, tile_world=world_for_tiles)
def run(spp):
    if not args.batch:
        for s in range(spp):
            mb.sample(s)
        return
    K, b = args.batch, 0
    for s0 in range(0, spp, K):
        ss = list(range(s0, min(spp, s0 + K)))
        g = b % args.groups
        mb.sample_batch(ss, lanes=mb.lanes[g * K: g * K + len(ss)])
        b += 1


run(len(mb.lanes))
mb.rays_traced(reset=True)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
    torch.cuda.synchronize()
import gc
gc.collect(); gc.disable()      # a cyclic collection inside the timed region is a 40 ms host stall in one sample_batch call (seen at random calls: 10 % of a 64-spp run)
t0 = time.perf_counter()
run(args.spp)
t_enqueued = time.perf_counter() - t0
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
rays = mb.rays_traced()
if world > 1:
    dev = torch.device("cpu") if rehearse else torch.device("cuda", local_rank)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    r = torch.tensor([rays], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(r)
    dt, rays = float(t[0]), int(r[0])
    dist.destroy_process_group()
    if rank != 0:
        sys.exit(0)
print(json.dumps({"metric": "Mrays/s", "value": round(rays / dt / 1e6, 1), "unit": "Mrays/s", "n_gpus": world, "scaling": "strong",
                  **({"REHEARSAL_not_a_measurement": "all ranks on cuda:0, gloo"} if rehearse else {}),
                  "config": {"workload": f"C5: atrium {len(sc['triangles'])} triangles, {W}x{H}, {args.spp} spp, depth <= {args.depth}, "
                                         "primary + (shadow + mirror|ambient continuation) per bounce", "rays": rays,
                             "rays_per_pixel_sample": round(rays / (n * args.spp), 3), "samples_in_flight": len(mb.lanes), "samples_per_trace_launch": args.batch or 1,
                             "chain": "rtamd_multibounce_sample_batch_device (C entry point)" if args.batch else "issued call by call from raytracer_amd/multibounce.py",
                             "parallelism": f"tiles{world}" if world > 1 else (f"rank 0's share of tiles{args.share_of}, alone on one GPU" if args.share_of > 1 else "single")},
                  "seconds": round(dt, 4), "host_enqueue_seconds": round(t_enqueued, 4), "ms_per_sample_frame": round(dt / args.spp * 1e3, 3), "data": "synthetic", "dtype": "f32"}))
