#!/usr/bin/env python3
"""bench.py — headline benchmark: Mrays/s and ms/frame at 1080p on the
Sponza-class scene (BASELINE.json configs[2]), 1..N MI355X.

A step is one frame of the fused hot path (ray_trace_and_shade.frag semantics):
for every pixel one primary closest-hit traversal, for every hit pixel normal /
uv interpolation, one sun shadow traversal, one bilinear texture fetch and the
Phong/sky shading, RGBA32F out.  Rays (dir/org, 24 B per pixel), scene and
textures are resident in HBM before the timed region.

Frames are submitted in batches (rtamd_render_batch_device, default 8 frames per
launch on one GPU): the persistent kernel pulls the work items of all frames of a
batch from one queue, so the slow tail of a frame overlaps the next frame's items —
how a real-time renderer keeps frames in flight — with ONE launch per batch on ONE
stream and the box's default hardware-queue count.  Launches do not overlap, so the
HIP-event duration of a launch divided by its frames IS the time per frame that the
roofline figures use.  `--batch 1` is the strictly serial number (one frame per launch).

With N > 1 ranks the frame is split into interleaved 32x32 tiles (scene replicated):
every rank renders its tiles of G consecutive frames with one launch (RGBA32F, as on one
GPU), converts them with the display pass (util/display.frag: float -> 8-bit back
buffer), one RCCL gather per G frames brings the RGBA8 shares to rank 0, which assembles
the scan-line frames — all inside the timed steps; two batch buffers alternate, so the
gather of one batch runs beside the rendering of the next.  (--gather-format f32 moves
the float shares instead: 33 MB per frame is 138 GB/s at this frame rate, more than an
xGMI link carries.)

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# The HIP runtime reads GPU_MAX_HW_QUEUES once, when it starts.  The bench leaves the box's value alone (batched launches
# need no extra hardware queues); RTAMD_BENCH_HW_QUEUES=<n> sets it for experiments with several streams.
_hwq = os.environ.get("RTAMD_BENCH_HW_QUEUES", "0")
if _hwq != "0":
    os.environ["GPU_MAX_HW_QUEUES"] = str(max(1, min(32, int(_hwq))))

# Peaks (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E 8 TB/s; fp32 vector 157.3 TFLOP/s counts an FMA as two, and the
# path may not fuse (bit-exactness), so 78.6 T un-fused op/s; one wave64 VALU instruction issues in 2 cycles per SIMD-32:
# 1024 SIMDs x 2.4 GHz / 2 = 1228.8 G wave-instructions/s (tools/micro/valu_ops.hip sustains 1.0 ns per instruction
# per SIMD = 1024 G/s under load); the L1 (TCP) looks up one cache line per clock and CU: 256 x 2.4 GHz = 614.4 G/s.
HBM_PEAK_GBS = 8000.0
VALU_PEAK_TOPS = 78.6
VALU_ISSUE_PEAK_GINST = 1228.8
TCP_ACCESS_PEAK_G = 614.4


WORKLOAD_NOTE = {"f16_640x480": "res/models/f16.obj + F16s.bmp / F16t.bmp (BASELINE config 1, the reference's default scene)",
                 "atrium_1080p": "procedural Sponza-class atrium", "atrium_small": "procedural Sponza-class atrium",
                 "soup_10m_4k": "seeded triangle soup (BASELINE config 4)", "soup_1m_4k": "seeded triangle soup",
                 "bunny_1080p": "bunny subdivided twice (BASELINE config 2)"}


class _FixtureModel:
    """A committed fixture's scene as uploaded (BVH already built by the host layer when the fixture was made)."""

    def __init__(self, g):
        self._sc = {"nodes": g["nodes"], "triangles": g["triangles"], "positions": g["positions"],
                    "tex_coords": g["tex_coords"] if "tex_coords" in g else None, "normals": g["normals"] if "normals" in g else None}

    def scene_arrays(self):
        return self._sc

    def has_tex_coords(self):
        return self._sc["tex_coords"] is not None

    def has_normals(self):
        return self._sc["normals"] is not None

    def build_bvh(self, threads=1):
        pass


def build_workload(name):
    from raytracer_amd import host, scenes
    if name == "f16_640x480":
        # BASELINE config 1: the reference's default scene (src/main.rs:30, f16.obj with its two BMP textures) from the
        # committed fixture, at the config's 640x480, with the fixture's camera (the 160x120 llvmpipe frame is this
        # camera at a quarter of the resolution: tests/test_host_layer.py checks the rays)
        z = np.load(os.path.join(REPO, "tests", "golden", "golden_f16.npz"), allow_pickle=False)
        model = _FixtureModel(z)
        w, h = 640, 480
        eye, d, fov = scenes.framing_camera(z["positions"], 0.55)
        return model, host.Camera(w / h, fov=fov, position=eye, direction=d), w, h, (z["tex0"], z["tex1"])
    if name == "atrium_1080p":
        tri, pos, uv, nrm, cam = scenes.atrium()
        model = host.Model.from_arrays(tri, pos, uv, nrm)
        w, h = 1920, 1080
        tex = (scenes.checker_texture(), scenes.noise_texture())
    elif name == "atrium_small":
        tri, pos, uv, nrm, cam = scenes.atrium()
        model = host.Model.from_arrays(tri, pos, uv, nrm)
        w, h = 480, 270
        tex = (scenes.checker_texture(256), scenes.noise_texture(128, 64))
    elif name == "soup_10m_4k":   # BASELINE config 4 (single-GPU share of it): not the default bench line
        tri, pos, cam = scenes.triangle_soup(10_000_000)
        model = host.Model.from_arrays(tri, pos)
        w, h = 3840, 2160
        tex = (None, None)
    elif name == "bunny_1080p":   # BASELINE config 2: bunny (from the committed fixture) subdivided twice, white albedo
        z = np.load(os.path.join(REPO, "tests", "golden", "golden_bunny.npz"), allow_pickle=False)
        t = z["triangles"]
        tri, pos, nrm = scenes.subdivide(np.stack([t["p0"], t["p1"], t["p2"], t["mat_idx"]], 1), z["positions"], z["normals"], levels=2)
        model = host.Model.from_arrays(tri, pos, None, nrm)
        w, h = 1920, 1080
        tex = (None, None)
        lo, hi = pos.min(0), pos.max(0)
        c = (lo + hi) / 2
        eye = c + 0.9 * np.linalg.norm(hi - lo) / 2 * np.float32([0.7, 0.3, 0.65])
        cam = {"position": tuple(float(x) for x in eye), "direction": tuple(float(x) for x in (c - eye)), "fov_deg": 60.0}
    elif name == "soup_1m_4k":
        tri, pos, cam = scenes.triangle_soup(1_000_000)
        model = host.Model.from_arrays(tri, pos)
        w, h = 3840, 2160
        tex = (None, None)
    else:
        raise SystemExit(f"unknown workload {name}")
    model.build_bvh(min(16, os.cpu_count() or 1))   # BVHBuilder::build_parallel: the serial build's tree, bit for bit
    d = np.asarray(cam["direction"], np.float32)
    d /= np.linalg.norm(d)
    camera = host.Camera(w / h, fov=math.radians(cam["fov_deg"]), position=cam["position"], direction=d)
    return model, camera, w, h, tex


def gather_to_root(send, gathered, rank, world, through_host=False):
    """The one collective of the N > 1 path: every rank's share `send` ([n] or [n, 4], on the GPU) lands in row `rank` of
    rank 0's `gathered` ([world, n(, 4)]), enqueued on torch's current stream (RCCL; xGMI between peers).  The rows of
    `gathered` are handed over as views, so the shares arrive in place and the assembly kernel reads them where they landed.
    through_host: the one-GPU rehearsal (gloo, all ranks on cuda:0) moves the shares through host memory instead."""
    import torch
    import torch.distributed as dist
    if through_host:
        mine = send.cpu()
        parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, gather_list=parts, dst=0)
        if rank == 0:
            gathered.copy_(torch.stack(parts))
    elif rank == 0:
        dist.gather(send, gather_list=list(gathered.unbind(0)), dst=0)
    else:
        dist.gather(send, dst=0)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher's environment: start N fresh rank processes (one per GPU) with
    torch.distributed.run on this node and return its exit status; rank 0 prints the JSON line.  Nothing in THIS process
    touches the GPU (device_count() only counts), so the ranks start from a clean runtime.  Fewer than N GPUs is an error,
    never a silent one-GPU run (RTAMD_BENCH_REHEARSE=1, the one-GPU rehearsal of the N > 1 code path, is exempt)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and os.environ.get("RTAMD_BENCH_REHEARSE") != "1":
        print(f"bench.py: --gpus {n} asked for, but this node shows {have} GPU(s); not falling back to fewer", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="atrium_1080p")
    ap.add_argument("--variant", type=int, default=0, help="0 persistent (default), 1 static A/B baseline (always one frame per launch)")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per launch (rtamd_render_batch_device), 1..16; 1 = one frame at a time.  0 = auto: 8 on one GPU "
                         "(4 for runs under 16 steps); with N > 1 the frames of one gather")
    ap.add_argument("--streams", type=int, default=0,
                    help="one GPU: batches alternate over that many streams (0 = auto: 2, so that the tail of one batch runs beside the "
                         "start of the next; 1: launches never overlap)")
    ap.add_argument("--gather-batch", type=int, default=0,
                    help="N > 1: frames per RCCL gather = frames per launch (0 = auto: 8 for 2-4 ranks, 16 beyond at up to 1080p; 8 for larger frames)")
    ap.add_argument("--gather-format", choices=("f32", "rgba8"), default="rgba8",
                    help="N > 1: what crosses xGMI. rgba8 (default) = the display pass's 8-bit back buffer (util/display.frag) that "
                         "every rank produces from its RGBA32F share; f32 = the RGBA32F shares themselves: 33 MB per 1080p frame, "
                         "i.e. 138 GB/s at this frame rate — more than an xGMI link carries")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="rtamd_set_option on the context before anything is launched (A/B runs; repeatable); recorded in config.options")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra measurements after the timed region (one frame at a time, non-overlapping batches): "
                         "for profiler passes, where every dispatch of the kernel should hold the same number of frames")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand without a launcher: start the N ranks ourselves and hand their output (rank 0's line) through
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from raytracer_amd.renderer import TILE_PIXELS, HipRayTracer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # Rehearsal on a one-GPU box (not a measurement): RTAMD_BENCH_REHEARSE=1 puts every rank on
    # cuda:0 and moves the gather through gloo/host memory, to exercise the N>1 stream logic.
    rehearse = world > 1 and os.environ.get("RTAMD_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    t0 = time.time()
    model, camera, W, H, tex = build_workload(args.workload)
    sc = model.scene_arrays()
    build_s = time.time() - t0
    has_tc, has_n = model.has_tex_coords(), model.has_normals()

    tracer = HipRayTracer(local_rank)
    tracer.set_kernel_variant(args.variant)
    for kv in args.option:
        name, _, value = kv.partition("=")
        tracer.set_option(name, int(value))
    tracer.upload_scene(sc)
    for slot, t in enumerate(tex):
        if t is not None:
            tracer.upload_texture(slot, t)
    # a non-default stream shared by torch (events, RCCL) and the library's launches
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    tracer.set_stream(stream.cuda_stream)

    # rays resident in HBM (the fused shader's `dir` / `org` textures)
    n_px = W * H
    inv = camera.inv_proj_view()
    d_dir = torch.empty((n_px, 3), dtype=torch.float32, device=dev)
    tracer.generate_rays_device(W, H, inv, d_dir.data_ptr())
    tracer.synchronize()
    d_org = torch.tensor(np.asarray(camera.position, np.float32), device=dev).repeat(n_px, 1).contiguous()

    per_rank_tiles = tracer.tiles_per_rank(W, H, world)
    n_local_out = n_px if world == 1 else per_rank_tiles * TILE_PIXELS
    # Frames per launch.  One GPU: 8.  N > 1: the frames of one gather (one collective per G frames: the call costs the root's
    # host 13+ us, a third of the time a 1/8 share of this frame takes; and a launch over one share of ONE frame is bounded by its
    # slowest pixel, not by throughput).
    if world > 1:
        # measured on one GPU (tools/share_cost.py: a rank's 1/8 share of this frame, render + display pass, two alternating streams):
        # 134 us per frame with one frame per launch, 52 us with 4, 37 us with 8
        # (16 per launch: 33 us).  Auto: a launch of about two whole frames' work — 8 frames up to 4 ranks, 16 beyond; 8 for larger
        # frames (10 M-triangle soup at 4K, a 1/8 share: 786 us per frame with one frame per launch, 550 with 4, 513 with 8)
        B = args.gather_batch if args.gather_batch > 0 else (args.batch if args.batch > 0 else ((8 if world <= 4 else 16) if n_px <= 1920 * 1080 else 8))
    else:
        # 8 frames per launch (a 20-step run: 8 + 8 + 4, measured equal to five launches of 4 — 0.240 ms per frame — with the
        # per-launch figures of the steady state); 4 only for runs shorter than two such launches
        # (round 2 used 2 frames per launch for scenes beyond the 256 MB Infinity Cache; re-measured in round 3 the 10 M soup too is fastest
        # with 8: 3.683 / 3.638 / 3.623 / 3.616 ms per frame with 2 / 4 / 6 / 8)
        B = args.batch if args.batch > 0 else (8 if args.steps >= 16 else 4)
    B = max(1, min(16, B))
    if args.variant != 0:
        B = 1
    # Batch j = frames jB .. jB+B-1 renders with one launch on render_streams[j % S] into batch buffer j % S ([B][n_local_out]
    # pixels).  N > 1: `stream` then waits for that stream, converts (rgba8), gathers the buffer, assembles the B scan-line
    # frames and marks the buffer free, while the next batch renders on the other stream.
    S = (max(1, min(4, args.streams)) if args.streams > 0 else (2 if args.variant == 0 else 1)) if world == 1 else 2
    # N > 1: two render streams of their own, so that `stream` (gather + assembly) never sits between two renders
    render_streams = ([stream] + [torch.cuda.Stream(device=dev) for _ in range(S - 1)]) if world == 1 else [torch.cuda.Stream(device=dev) for _ in range(S)]
    rgba = [torch.zeros((B * n_local_out, 4), dtype=torch.float32, device=dev) for _ in range(S)]
    consumed = [torch.cuda.Event() for _ in range(S)]   # batch buffer f has been gathered
    d_rgba = rgba[0]
    frame = None
    as8 = world > 1 and args.gather_format == "rgba8"
    send = rgba
    if as8:   # each rank also writes its share as the display pass would (8-bit back buffer); the gather moves that
        send = [torch.zeros((B * n_local_out,), dtype=torch.int32, device=dev) for _ in range(S)]
    if world > 1 and rank == 0:
        if as8:
            gathered = torch.empty((world, B * n_local_out), dtype=torch.int32, device=dev)
            frames_out = torch.empty((B, n_px), dtype=torch.int32, device=dev)
        else:
            gathered = torch.empty((world, B * n_local_out, 4), dtype=torch.float32, device=dev)   # [rank][frame of the batch][pixel]
            frames_out = torch.empty((B, n_px, 4), dtype=torch.float32, device=dev)
        frame = frames_out[0]
    launch_events = []     # (frames in the launch, start event, stop event)
    def launch(f, n_frames, timed):
        """Frames 0..n_frames-1 of batch buffer f in ONE launch on render_streams[f]."""
        rs = render_streams[f]
        tracer.set_stream(rs.cuda_stream)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(rs)
        kw = dict(has_tex_coords=has_tc, has_normals=has_n, tile_rank=rank, tile_world=world)
        if n_frames == 1:
            tracer.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), rgba_ptr=rgba[f].data_ptr(), **kw)
        else:
            tracer.render_batch_device(W, H, [{"dir_ptr": d_dir.data_ptr(), "org_ptr": d_org.data_ptr(), "rgba_ptr": rgba[f].data_ptr() + g * n_local_out * 16}
                                              for g in range(n_frames)], **kw)
        if timed:
            e1.record(rs)
            launch_events.append((n_frames, e0, e1))
        if as8:   # the display pass over the whole batch buffer, behind the render on the same stream
            tracer.display_rgba8_device(rgba[f].data_ptr(), n_frames * n_local_out, send[f].data_ptr())

    gather_events = []     # rank 0, timed region: (share ready, gather done, assembly done) on `stream`, one triple per batch
    def flush(f, timed=False):
        """Gather batch buffer f (all ranks call this at the same step) and assemble its frames on the root."""
        rs = render_streams[f]
        if rs is not stream:
            stream.wait_stream(rs)
        if rehearse:
            stream.synchronize()
        timed = timed and rank == 0
        if timed:
            g0, g1, a1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            g0.record(stream)       # behind the wait: this rank's own share is ready
        gather_to_root(send[f], gathered if rank == 0 else None, rank, world, rehearse)
        if timed:
            g1.record(stream)       # torch makes `stream` wait for the collective: every peer's share has landed
        if rank == 0:
            tracer.set_stream(stream.cuda_stream)
            tracer.assemble_tile_batch_device(gathered.data_ptr(), W, H, world, 1 if as8 else 4, B, frames_out.data_ptr())
        if timed:
            a1.record(stream)
            gather_events.append((g0, g1, a1))
        consumed[f].record(stream)

    def run(n_steps, timed=False):
        """n_steps frames: full batches of B, then one smaller batch for the rest; every frame is rendered (and, N > 1,
        gathered and assembled) by the time the streams are joined."""
        done, batch = 0, 0
        while done < n_steps:
            nf = min(B, n_steps - done)
            f = batch % S
            if world > 1 and batch >= S:
                render_streams[f].wait_event(consumed[f])      # the buffer is free again once batch - S has been gathered
            launch(f, nf, timed)
            if world > 1:
                flush(f, timed)
            done += nf
            batch += 1
        for rs in render_streams:
            if rs is not stream:
                stream.wait_stream(rs)

    # ---- untimed: per-pixel counters of the whole frame (rank-local share) for the roofline
    d_counts = torch.zeros((n_local_out, 2), dtype=torch.int32, device=dev)
    d_hit = torch.zeros((n_local_out,), dtype=torch.int32, device=dev)
    tracer.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=has_tc, has_normals=has_n,
                         tile_rank=rank, tile_world=world, counts_ptr=d_counts.data_ptr(), hit_id_ptr=d_hit.data_ptr(),
                         rgba_ptr=d_rgba.data_ptr())
    torch.cuda.synchronize()
    local_tiles = tracer.local_tile_count(W, H, rank, world)
    if world == 1:
        n_primary = n_px
        valid = torch.ones((n_px,), dtype=torch.bool, device=dev)
    else:
        # compact tile layout: pixels of edge tiles outside the frame are never written and carry no rays
        valid = (d_counts[: local_tiles * TILE_PIXELS, 0] > 0)
        n_primary = int(valid.sum().item())
        valid = torch.cat([valid, torch.zeros(n_local_out - valid.numel(), dtype=torch.bool, device=dev)])
    hitmask = (d_hit != -1) & valid
    n_shadow = int(hitmask.sum().item())
    n_box = int(d_counts[:, 0][valid].to(torch.int64).sum().item())
    n_tri = int(d_counts[:, 1][valid].to(torch.int64).sum().item())
    stats = torch.tensor([n_primary, n_shadow, n_box, n_tri], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(stats)
    n_primary_all, n_shadow_all, n_box_all, n_tri_all = (int(x) for x in stats.tolist())
    rays_per_frame = n_primary_all + n_shadow_all

    # Python's cyclic collector stays out of the timed region (in tools/bench_c5.py a collection stalled one host call for 40 ms).  Collected
    # HERE, before the last pieces of set-up load the GPU: a pause just before the timed region would let the clocks drop (a 20-step run
    # measured 0.280 instead of 0.244 ms per frame with the collection placed there).
    import gc
    gc.collect()
    gc.disable()

    # ---- untimed: pick the partner stream.  This is the last piece of set-up on purpose: it renders about a hundred frames, so the
    # warm-up that follows starts from a loaded GPU (a 20-step run after an idle gap measures the clock ramp: 0.261 ms per frame against
    # 0.249 after 50 warm-up frames and 0.245 after 500, same code)
    if S == 2 and args.variant == 0:
        # two streams that share a hardware queue cannot overlap (HIP spreads a process's streams over GPU_MAX_HW_QUEUES
        # queues): try a handful of partner streams and keep the one that overlaps best with the first render stream
        # (every rank for itself, on its own tile share: no collective here)
        def pair_ms(partner, batches=6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(batches):
                tracer.set_stream((render_streams[0], partner)[k % 2].cuda_stream)
                tracer.render_batch_device(W, H, [{"dir_ptr": d_dir.data_ptr(), "org_ptr": d_org.data_ptr(), "rgba_ptr": rgba[k % 2].data_ptr() + g * n_local_out * 16}
                                                  for g in range(B)], has_tex_coords=has_tc, has_normals=has_n, tile_rank=rank, tile_world=world)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / batches
        candidates = [render_streams[1]] + [torch.cuda.Stream(device=dev) for _ in range(3)]
        for c in candidates:
            pair_ms(c, 2)   # first use of a stream creates its queue
        render_streams[1] = min(candidates, key=pair_ms)
        tracer.set_stream(stream.cuda_stream)

    # ---- warmup + timed region
    tracer.set_stream(stream.cuda_stream)
    torch.cuda.synchronize()
    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record(stream)
    run(args.steps, timed=True)
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    gc.enable()
    ev_ms = ev0.elapsed_time(ev1)
    tt = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, ev_ms = float(tt[0]), float(tt[1])

    # ---- duration of the dominant kernel's launches: HIP events on the launch stream around every launch.  In the timed
    # region batches alternate over S streams and their launches overlap, so those spans say nothing about one launch
    # (they are kept as `overlapped_launch_ms`); the figure the roofline uses comes from launches that do NOT overlap:
    # after the timed region, the same batches back to back on one stream (what `--streams 1` runs, and what the committed
    # rocprofv3 --kernel-trace --stats summary of `--streams 1` shows as the kernel's average duration).
    full = [e0.elapsed_time(e1) for nf, e0, e1 in launch_events if nf == B]
    if not full:
        full = [e0.elapsed_time(e1) * B / nf for nf, e0, e1 in launch_events]
    overlapped_launch_ms = float(np.mean(full))
    launch_ms = overlapped_launch_ms
    n_alone = 0
    if S > 1 and world == 1 and not args.no_extras:
        launch_events.clear()
        n_alone = max(5, min(60, args.steps // B))
        torch.cuda.synchronize()
        for _ in range(n_alone):
            launch(0, B, True)
        torch.cuda.synchronize()
        launch_ms = float(np.mean([e0.elapsed_time(e1) for nf, e0, e1 in launch_events]))
    frame_ms_in_launch = launch_ms / B
    tracer.set_stream(stream.cuda_stream)

    # ---- untimed extra for the record: the same frame one at a time (what a caller that waits for every frame sees)
    serial_ms = None
    if world == 1 and B > 1 and not args.no_extras:
        n_serial = max(10, min(200, args.steps))
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(stream)
        for _ in range(n_serial):
            tracer.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=has_tc, has_normals=has_n, rgba_ptr=rgba[0].data_ptr())
        s1.record(stream)
        torch.cuda.synchronize()
        serial_ms = s0.elapsed_time(s1) / n_serial

    comm = None
    if world > 1:
        # the collective and the root's assembly reported apart from the rendering.  Spans on the root's gather
        # stream (HIP events): "gather" starts when the root's own share is ready and ends when every peer's share has
        # landed, so it includes waiting for the slowest peer's render, not only the transfer.
        comm = {"backend": dist.get_backend(), "ranks": dist.get_world_size()}
        if rank == 0 and gather_events:
            comm["gather_ms_per_frame"] = round(sum(g0.elapsed_time(g1) for g0, g1, _ in gather_events) / args.steps, 5)
            comm["assemble_ms_per_frame"] = round(sum(g1.elapsed_time(a1) for _, g1, a1 in gather_events) / args.steps, 5)
            comm["bytes_per_peer_per_frame"] = n_local_out * (4 if as8 else 16)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    frame_check = None
    if world > 1:
        # untimed: the assembled frames of the last batch must equal this rank rendering the whole frame alone
        whole = torch.zeros((n_px, 4), dtype=torch.float32, device=dev)
        tracer.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=has_tc, has_normals=has_n, rgba_ptr=whole.data_ptr())
        torch.cuda.synchronize()
        n_last = (args.steps - 1) % B + 1      # frames the last gathered batch holds
        if as8:
            whole8 = torch.zeros((n_px,), dtype=torch.int32, device=dev)
            tracer.display_rgba8_device(whole.data_ptr(), n_px, whole8.data_ptr())
            torch.cuda.synchronize()
            whole = whole8
        frame_check = all(bool(torch.equal(whole.view(torch.int32).reshape(-1), frames_out[i].view(torch.int32).reshape(-1))) for i in range(n_last))

    ms_per_step = elapsed * 1e3 / args.steps
    value = rays_per_frame * args.steps / elapsed / 1e6

    # algorithmic bytes / ops of ONE frame on this rank:
    #   36 B per AABB test + 52 B per triangle test + 24 B ray in per ray + 16 B RGBA32F out per pixel
    #   + per shaded pixel 3x12 B normals + 3x8 B uvs + 16 B triangle record + 4 texels x 4 B
    shaded_extra = (36 if has_n else 0) + (24 if has_tc else 0) + 16 + 16
    alg_bytes = 36 * n_box + 52 * n_tri + 24 * (n_primary + n_shadow) + 16 * n_primary + shaded_extra * n_shadow
    alg_ops = 25 * n_box + 60 * n_tri + 150 * n_shadow
    frame_s = frame_ms_in_launch * 1e-3
    alg_gbs = alg_bytes / frame_s / 1e9
    valu_tops = alg_ops / frame_s / 1e12

    # Counters of the same kernel on the same workload from the committed rocprofv3 --pmc passes (PMC runs serialise
    # kernels, so they cannot be collected inside the timed region); the record names the roof that bounds the workload.
    cpath = os.path.join(REPO, "profiles", f"counters_{args.workload}.json")
    prof = json.load(open(cpath)) if world == 1 and args.variant == 0 and os.path.exists(cpath) else None
    roof = {"kernel": "k_persistent<fused>" if args.variant == 0 else "k_static<fused>",
            "launch": {"frames": B, "ms": round(launch_ms, 5), "ms_per_frame": round(frame_ms_in_launch, 5), "launches_overlap": False if n_alone else S > 1,
                       "timing": (f"HIP events on the launch stream around {n_alone} launches of {B} frames, back to back on ONE stream after the timed "
                                  f"region (as `--streams 1`); in the timed region batches alternate over {S} streams and a launch's span is "
                                  f"{overlapped_launch_ms:.4f} ms beside its neighbour") if n_alone else
                                 "HIP events on the launch stream around every timed launch of a full batch"},
            "frac_at_job_rate_note": "every frac here divides by launch.ms_per_frame; at the job's rate multiply by launch.ms_per_frame / ms_per_step",
            "stream_ms_per_step_hip_events": round(ev_ms / args.steps, 5),
            "aabb_tests_per_ray": round(n_box / max(1, n_primary + n_shadow), 2),
            "tri_tests_per_ray": round(n_tri / max(1, n_primary + n_shadow), 2),
            "valu": {"algorithmic_ops_per_frame": alg_ops, "achieved_Tops": round(valu_tops, 3), "peak_Tops": VALU_PEAK_TOPS,
                     "frac": round(valu_tops / VALU_PEAK_TOPS, 5), "note": "SURVEY 8d operation count (25 per AABB test, 60 per triangle test, 150 per shaded "
                     "pixel) over the un-fused fp32 peak"},
            "hbm": {"algorithmic_bytes_per_frame": alg_bytes, "algorithmic_GBs": round(alg_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s"}}
    if prof:
        pf = prof["per_frame"]
        traffic = pf["hbm_bytes"]
        hbm_gbs = traffic / frame_s / 1e9
        roof["hbm"].update({"traffic_bytes_per_frame": traffic, "achieved": round(hbm_gbs, 2), "frac": round(hbm_gbs / HBM_PEAK_GBS, 5),
                            "algorithmic_over_traffic": round(alg_bytes / traffic, 2),
                            "cache_served": alg_bytes > 2 * traffic,
                            "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE of the PMC passes (gfx950 tallies the 128-B requests of wide loads at "
                                    "64 B); the algorithmic bytes are what the reference's layout would move per test: where they exceed the "
                                    "traffic, L1 / L2 / Infinity Cache serve them"})
        inst = pf["valu_wave_instructions"]
        ginst = inst / frame_s / 1e9
        roof["valu"].update({"wave_instructions_per_frame": inst, "issue_Ginst_s": round(ginst, 2), "issue_peak_Ginst_s": VALU_ISSUE_PEAK_GINST,
                             "issue_frac": round(ginst / VALU_ISSUE_PEAK_GINST, 5)})
        tcp = pf.get("tcp_cache_accesses")
        if tcp:
            roof["l1"] = {"tcp_cache_accesses_per_frame": tcp, "achieved_G_s": round(tcp / frame_s / 1e9, 2), "peak_G_s": TCP_ACCESS_PEAK_G,
                          "frac": round(tcp / frame_s / 1e9 / TCP_ACCESS_PEAK_G, 5)}
        if "l1_path_busy" in prof:
            roof.setdefault("l1", {})["path_busy_frac"] = round(prof["l1_path_busy"]["ta_busy_frac"], 4)
            roof["l1"]["path_busy_note"] = "TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8) of one PMC pass: share of the kernel's cycles in which a CU's texture-address unit (the front of the L1 path) is busy"
        roof["counters_from"] = prof["source"]
        if prof.get("bound") == "tcp_access_rate" and tcp:
            limiter = {"resource": "l1_access_rate", "achieved": roof["l1"]["achieved_G_s"], "peak": TCP_ACCESS_PEAK_G, "unit": "G cache-line look-ups/s",
                       "frac": roof["l1"]["frac"],
                       "frac_note": "TCP_TOTAL_CACHE_ACCESSES per second over one look-up per clock and CU at the 2.4 GHz maximum clock; a 16-byte lane access "
                                    "counts two look-ups, 8- and 4-byte ones one, and the narrow ones can share a cycle, so a kernel AT the L1's limit reads a few "
                                    "per cent above 1 (tools/micro/tcp_rate.hip sustains 2.38 look-ups per ns and CU with 16-byte accesses only)"}
        else:
            limiter = {"resource": "valu_issue", "achieved": round(ginst, 2), "peak": VALU_ISSUE_PEAK_GINST, "unit": "G wave64 VALU instructions/s",
                       "frac": roof["valu"]["issue_frac"],
                       "frac_note": "issued wave64 VALU instructions (SQ_INSTS_VALU, masked lanes and bookkeeping included) over 1024 SIMDs x 2.4 GHz / 2: "
                                    "how busy the issue slots are, not how much of it is the algorithm's arithmetic - that is roofline.frac"}
        traffic_per_launch = traffic * B      # HBM bytes per launch
    else:
        limiter, traffic_per_launch = None, None
    # The headline fraction is SURVEY 8d's: the path's ALGORITHMIC operations (25 per AABB test, 60 per triangle test, 150 per shaded
    # pixel, from the kernel's own per-pixel test counters) per second over the un-fused fp32 vector peak.  The scene is cache-served
    # (algorithmic bytes exceed the HBM traffic many times over), so the byte formula is reported under roofline.hbm, not priced here.
    head = {"bound": "valu", "achieved": round(valu_tops, 3), "peak": VALU_PEAK_TOPS, "unit": "Top/s (un-fused fp32)",
            "frac": round(valu_tops / VALU_PEAK_TOPS, 5), "traffic": traffic_per_launch,
            **({"issue_frac": roof["valu"]["issue_frac"], "limiter": limiter} if limiter else {})}
    out = {
        "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {WORKLOAD_NOTE.get(args.workload, 'scene')}, {len(sc['triangles'])} triangles, "
                               f"{W}x{H}, 1 spp, primary + sun shadow ray + shading (fused ray_trace_and_shade semantics)",
                   "rays_per_frame": rays_per_frame, "primary_rays": n_primary_all, "shadow_rays": n_shadow_all,
                   "parallelism": f"tiles{world}" if world > 1 else "single", "kernel_variant": args.variant,
                   "frames_per_launch": B, "streams": S, **({"options": args.option} if args.option else {}),
                   **({"gather_format": args.gather_format, "rccl_ranks": comm["ranks"] if comm["backend"] == "nccl" else 0,
                       "collective_backend": comm["backend"], "gather_ms_per_frame": comm.get("gather_ms_per_frame"),
                       "assemble_ms_per_frame": comm.get("assemble_ms_per_frame"), "gather_bytes_per_peer_per_frame": comm.get("bytes_per_peer_per_frame"),
                       "gather_timing": "HIP events on rank 0's gather stream per batch: own share ready -> all shares landed (includes waiting "
                                        "for the slowest peer's render) -> scan-line frames assembled; both overlap the next batch's rendering"} if world > 1 else {}),
                   **({"ms_per_frame_one_at_a_time": round(serial_ms, 5)} if serial_ms is not None else {}), "shadow_rays_stop_at_first_occluder": True,
                   **({"assembled_frame_equals_single_gpu_frame": frame_check} if frame_check is not None else {}),
                   **({"REHEARSAL_not_a_measurement": "all ranks on cuda:0, gloo gather through host memory"} if rehearse else {}),
                   "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                   "bvh_build_s": round(build_s, 2)},
        "roofline": {**head, **roof},
    }

    if world == 1 and not args.no_cpu_baseline:
        # oracle A (the CPU restatement) on the same frame, bounded: every 2nd pixel row
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from oracles import OracleA
        A = OracleA()
        cores = min(16, os.cpu_count() or 1)
        step_rows = 2 if n_px >= 1920 * 1080 else 1
        dirs = d_dir.view(H, W, 3)[::step_rows].contiguous().cpu().numpy().reshape(-1, 3)
        orgs = np.broadcast_to(np.asarray(camera.position, np.float32), dirs.shape).copy()
        reps = 12
        tc0 = time.perf_counter()
        for _ in range(reps):
            r = A.render(sc, dirs, orgs, has_tc, has_n, tex[0], tex[1], nthreads=cores)
        cpu_s = time.perf_counter() - tc0
        cpu_rays = reps * (len(dirs) + int((r["hit_id"] != 0xFFFFFFFF).sum()))
        out["cpu_baseline"] = {"value": round(cpu_rays / cpu_s / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                               "sample": f"oracle/oracle_a.c, {reps} passes over every {'2nd pixel row' if step_rows == 2 else 'pixel'} of the same frame "
                                         f"({len(dirs)} pixels per pass, {cpu_rays} rays in all, {cpu_s:.1f} s wall on {cores} threads)"}
        # baseline B1: the unmodified reference shader on Mesa llvmpipe.  The shader text does not travel to the
        # GPU box, so this is the committed record of the build container's run (tools/llvmpipe_baseline.py), not a live figure.
        lpath = os.path.join(REPO, "profiles", "llvmpipe_baseline.json")
        if os.path.exists(lpath):
            rec = json.load(open(lpath)).get(args.workload)
            if rec:
                out["cpu_baseline"]["reference_llvmpipe"] = {k: rec[k] for k in ("mrays_per_s", "ms_per_frame", "host_cores", "llvmpipe_threads",
                                                                                  "renderer", "where", "timing", "shader")}
                out["cpu_baseline"]["reference_llvmpipe"]["record"] = "profiles/llvmpipe_baseline.json"
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
