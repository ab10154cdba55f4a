#!/usr/bin/env python3
"""bench.py — headline benchmark: Mrays/s and ms/frame at 1080p on the
Sponza-class scene (BASELINE.json configs[2]), 1..N MI355X.

A step is one frame of the fused hot path (ray_trace_and_shade.frag semantics):
for every pixel one primary closest-hit traversal, for every hit pixel normal /
uv interpolation, one sun shadow traversal, one bilinear texture fetch and the
Phong/sky shading, RGBA32F out.  Rays (dir/org, 24 B per pixel), scene and
textures are resident in HBM before the timed region.  With N > 1 ranks the
frame is split into interleaved 32x32 tiles (scene replicated), every rank
renders its tiles (RGBA32F, as on one GPU) and converts them with the display pass
(util/display.frag: float -> 8-bit back buffer), one RCCL gather per 4 frames brings
the RGBA8 shares to rank 0, which assembles the scan-line frames — all inside the
timed steps.  (--gather-format f32 moves the float shares instead: 33 MB per frame is
138 GB/s at this frame rate, more than an xGMI link carries.)

Frames in flight (default 4, 8 with N > 1): consecutive frames are issued round-robin on HIP
streams (the library keeps per-stream launch scratch, include/rtamd.h), each into
its own output buffer, so the slow tail of frame k overlaps the start of frame
k+1 — how a real-time renderer submits frames.  All K frames complete inside the
timed region.  `--frames-in-flight 1` gives the strictly serial number.

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

# Frames in flight need hardware queues of their own: HIP spreads a process's streams over
# GPU_MAX_HW_QUEUES queues (default 4) and launches on streams that share one do not overlap.
# Measured (tools/launch_cost.py, one rank's 1/8 share of the frame): 65 us per frame with 4
# queues / 6 streams, 42 us with 16 queues / 8 streams.  Read by the HIP runtime when it starts;
# the GPU boxes export 4, so this replaces the value (RTAMD_BENCH_HW_QUEUES overrides, "0" keeps it).
_hwq = os.environ.get("RTAMD_BENCH_HW_QUEUES", "16")
if _hwq != "0":
    os.environ["GPU_MAX_HW_QUEUES"] = str(max(1, min(32, int(_hwq))))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
VALU_PEAK_TOPS = 78.6        # 157.3 TFLOP/s FMA-count
VALU_NS_PER_WAVE_INSTR = 1.0  # tools/micro/valu_ops.hip on MI355X: v_mul/v_sub/v_mov, and the visit's mix of classes
N_SIMDS = 1024               # 256 CUs x 4ed fp32 vector = 78.6 T un-fused op/s


WORKLOAD_NOTE = {"atrium_1080p": "procedural Sponza-class atrium", "atrium_small": "procedural Sponza-class atrium",
                 "soup_10m_4k": "seeded triangle soup (BASELINE config 4)", "soup_1m_4k": "seeded triangle soup",
                 "bunny_1080p": "bunny subdivided twice (BASELINE config 2)"}


def build_workload(name):
    from raytracer_amd import host, scenes
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="atrium_1080p")
    ap.add_argument("--variant", type=int, default=0, help="0 persistent (default), 1 static A/B baseline")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames issued round-robin on that many streams (1 = serial; 0 = auto: 4 on one GPU, 8 with tiles split over "
                         "several GPUs, where a launch covers 1/N of the frame and is bounded by its slowest rays, not by throughput)")
    ap.add_argument("--gather-batch", type=int, default=0,
                    help="N > 1: frames per RCCL gather (0 = auto: 4 up to 1080p, 1 above)")
    ap.add_argument("--gather-format", choices=("f32", "rgba8"), default="rgba8",
                    help="N > 1: what crosses xGMI. rgba8 (default) = the display pass's 8-bit back buffer (util/display.frag) that "
                         "every rank produces from its RGBA32F share; f32 = the RGBA32F shares themselves: 33 MB per 1080p frame, "
                         "i.e. 138 GB/s at this frame rate — more than an xGMI link carries")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd.renderer import TILE_PIXELS, HipRayTracer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    d_org = torch.tensor(np.asarray(camera.position, np.float32), device=dev).repeat(n_px, 1).contiguous()

    per_rank_tiles = tracer.tiles_per_rank(W, H, world)
    n_local_out = n_px if world == 1 else per_rank_tiles * TILE_PIXELS
    # auto: 4 frames in flight on one GPU (8 with N > 1, where a launch covers 1/N of the frame and is bounded by
    # its slowest rays); 2 when the scene records exceed the 256 MB Infinity Cache, where more frames in
    # flight only evict each other's nodes (10 M soup: 4.04 ms with 2, 4.30 ms with 4)
    info = tracer.scene_info()
    scene_mb = (info.inner_node_count * 64 + info.triangle_count * 48) / 1e6
    F = args.frames_in_flight if args.frames_in_flight > 0 else (2 if scene_mb > 256 else 4 if world == 1 else 8)
    F = max(1, min(8, F))
    # One GPU: frame k renders on render_streams[k % F] into rgba[k % F].
    # N > 1: frames are gathered in batches of G (one collective per G frames: the call costs the root's host
    # 13+ us, a third of the 39 us a 1/8 share of this frame takes).  Batch j = frames jG .. jG+G-1 renders on
    # render_streams[j % F], back to back, into batch buffer j % F ([G][n_local_out] pixels); `stream` then waits
    # for that stream, gathers the buffer, assembles the G scan-line frames and marks the buffer free.
    G = 1
    if world > 1:
        G = args.gather_batch if args.gather_batch > 0 else (4 if n_px <= 1920 * 1080 else 1)
    render_streams = [stream] + [torch.cuda.Stream(device=dev) for _ in range(F - 1)]
    rgba = [torch.zeros((G * n_local_out, 4), dtype=torch.float32, device=dev) for _ in range(F)]
    consumed = [torch.cuda.Event() for _ in range(F)]   # batch buffer f has been gathered
    d_rgba = rgba[0]
    if F == 2 and world == 1:
        # two streams that share a hardware queue cannot overlap: try a handful of partner streams and keep
        # the one that overlaps best (only matters when GPU_MAX_HW_QUEUES is left at a small value)
        def pair_ms(partner, frames=16):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(frames):
                tracer.set_stream((stream, partner)[k % 2].cuda_stream)
                tracer.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=has_tc, has_normals=has_n,
                                     tile_rank=rank, tile_world=world, rgba_ptr=rgba[k % 2].data_ptr())
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / frames
        candidates = [render_streams[1]] + [torch.cuda.Stream(device=dev) for _ in range(3)]
        for c in candidates:
            pair_ms(c, 4)   # first use of a stream creates its queue
        render_streams[1] = min(candidates, key=pair_ms)
        tracer.set_stream(stream.cuda_stream)
    frame = None
    as8 = world > 1 and args.gather_format == "rgba8"
    send = rgba
    if as8:   # each rank also writes its share as the display pass would (8-bit back buffer); the gather moves that
        send = [torch.zeros((G * n_local_out,), dtype=torch.int32, device=dev) for _ in range(F)]
    if world > 1 and rank == 0:
        if as8:
            gathered = torch.empty((world, G * n_local_out), dtype=torch.int32, device=dev)
            frames_out = torch.empty((G, n_px), dtype=torch.int32, device=dev)
        else:
            gathered = torch.empty((world, G * n_local_out, 4), dtype=torch.float32, device=dev)   # [rank][frame of the batch][pixel]
            frames_out = torch.empty((G, n_px, 4), dtype=torch.float32, device=dev)
        frame = frames_out[0]
    launch_events = []
    pending = [None]   # batch buffer that holds rendered frames not gathered yet

    def flush(f):
        """Gather batch buffer f (all ranks call this at the same step) and assemble its frames on the root."""
        rs = render_streams[f]
        if rs is not stream:
            stream.wait_stream(rs)
        if rehearse:
            stream.synchronize()
            mine = send[f].cpu()
            parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, gather_list=parts, dst=0)
            if rank == 0:
                gathered.copy_(torch.stack(parts))
        elif rank == 0:
            dist.gather(send[f], gather_list=list(gathered.unbind(0)), dst=0)
        else:
            dist.gather(send[f], dst=0)
        if rank == 0:
            tracer.set_stream(stream.cuda_stream)
            tracer.assemble_tile_batch_device(gathered.data_ptr(), W, H, world, 1 if as8 else 4, G, frames_out.data_ptr())
        consumed[f].record(stream)
        pending[0] = None

    def step(k, timed=False):
        batch, g = divmod(k, G)
        f = batch % F
        rs = render_streams[f]
        if world > 1 and g == 0 and batch >= F:
            rs.wait_event(consumed[f])          # the buffer is free again once batch - F has been gathered
        tracer.set_stream(rs.cuda_stream)
        # launch durations for the roofline: every launch on one GPU; with N > 1 the host has ~40 us per step
        # for everything (render call, gather, assembly), so only every 8th launch carries the two extra events
        timed = timed and (world == 1 or k % 8 == 0)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(rs)
        # with the RGBA8 gather the kernel writes the display-format pixel beside the float one (rtamd_frame_out.rgba8)
        tracer.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=has_tc, has_normals=has_n,
                             tile_rank=rank, tile_world=world, rgba_ptr=rgba[f].data_ptr() + g * n_local_out * 16,
                             rgba8_ptr=(send[f].data_ptr() + g * n_local_out * 4) if as8 else 0)
        if timed:
            e1.record(rs)
            launch_events.append((e0, e1))
        if world > 1:
            pending[0] = f
            if g == G - 1:
                flush(f)

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

    # ---- warmup + timed region
    tracer.set_stream(stream.cuda_stream)
    torch.cuda.synchronize()
    for k in range(args.warmup):
        step(k)
    if pending[0] is not None:
        flush(pending[0])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record(stream)
    for k in range(args.steps):
        step(k, timed=True)
    if pending[0] is not None:
        flush(pending[0])       # the last, partial batch: all K frames are gathered and assembled inside the timed region
    for rs in render_streams:
        if rs is not stream:
            stream.wait_stream(rs)
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    ev_ms = ev0.elapsed_time(ev1)
    tt = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, ev_ms = float(tt[0]), float(tt[1])

    # ---- duration of the dominant kernel's launches: HIP events on the launch stream around every
    # timed launch (with frames in flight the launches overlap, so this exceeds ms_per_step)
    kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in launch_events]))
    tracer.set_stream(stream.cuda_stream)

    # ---- untimed extra for the record: the same frame one at a time (what a caller that waits for every frame
    # sees); `value` and `ms_per_step` above are the pipelined figures
    serial_ms = None
    if world == 1 and F > 1:
        n_serial = max(10, min(200, args.steps))
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(stream)
        for _ in range(n_serial):
            tracer.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=has_tc, has_normals=has_n, rgba_ptr=rgba[0].data_ptr())
        s1.record(stream)
        torch.cuda.synchronize()
        serial_ms = s0.elapsed_time(s1) / n_serial

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    frame_check = None
    if world > 1:
        # untimed: the assembled frame of the last step must equal this rank rendering the whole frame alone
        full = torch.zeros((n_px, 4), dtype=torch.float32, device=dev)
        tracer.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=has_tc, has_normals=has_n, rgba_ptr=full.data_ptr())
        torch.cuda.synchronize()
        n_last = (args.steps - 1) % G + 1      # frames the last gathered batch holds
        if as8:
            full8 = torch.zeros((n_px,), dtype=torch.int32, device=dev)
            tracer.display_rgba8_device(full.data_ptr(), n_px, full8.data_ptr())
            torch.cuda.synchronize()
            full = full8
        frame_check = all(bool(torch.equal(full.view(torch.int32).reshape(-1), frames_out[i].view(torch.int32).reshape(-1))) for i in range(n_last))

    ms_per_step = elapsed * 1e3 / args.steps
    value = rays_per_frame * args.steps / elapsed / 1e6

    # algorithmic bytes / ops of ONE launch on this rank:
    #   36 B per AABB test + 52 B per triangle test + 24 B ray in per ray + 16 B RGBA32F out per pixel
    #   + per shaded pixel 3x12 B normals + 3x8 B uvs + 16 B triangle record + 4 texels x 4 B
    shaded_extra = (36 if has_n else 0) + (24 if has_tc else 0) + 16 + 16
    alg_bytes = 36 * n_box + 52 * n_tri + 24 * (n_primary + n_shadow) + 16 * n_primary + shaded_extra * n_shadow
    alg_ops = 25 * n_box + 60 * n_tri + 150 * n_shadow
    launch_ms = kernel_ms
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
    job_ms = ev_ms / args.steps
    out = {
        "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {WORKLOAD_NOTE.get(args.workload, 'scene')}, {len(sc['triangles'])} triangles, "
                               f"{W}x{H}, 1 spp, primary + sun shadow ray + shading (fused ray_trace_and_shade semantics)",
                   "rays_per_frame": rays_per_frame, "primary_rays": n_primary_all, "shadow_rays": n_shadow_all,
                   "parallelism": f"tiles{world}" if world > 1 else "single", "kernel_variant": args.variant,
                   "frames_in_flight": F, **({"frames_per_gather": G, "gather_format": args.gather_format} if world > 1 else {}),
                   **({"ms_per_frame_one_at_a_time": round(serial_ms, 5)} if serial_ms is not None else {}), "shadow_rays_stop_at_first_occluder": True,
                   **({"assembled_frame_equals_single_gpu_frame": frame_check} if frame_check is not None else {}),
                   **({"REHEARSAL_not_a_measurement": "all ranks on cuda:0, gloo gather through host memory"} if rehearse else {}),
                   "bvh_build_s": round(build_s, 2)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                     "kernel": "k_persistent<fused>" if args.variant == 0 else "k_static<fused>",
                     "launch_ms": round(launch_ms, 5), "launches_overlap": F > 1,
                     "stream_ms_per_step_hip_events": round(job_ms, 5),
                     "achieved_per_step": round(alg_bytes / (job_ms * 1e-3) / 1e9, 2),
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "aabb_tests_per_ray": round(n_box / max(1, n_primary + n_shadow), 2),
                     "tri_tests_per_ray": round(n_tri / max(1, n_primary + n_shadow), 2),
                     "valu": {"achieved_Tops": round(alg_ops / (launch_ms * 1e-3) / 1e12, 3), "peak_Tops": VALU_PEAK_TOPS,
                              "frac": round(alg_ops / (launch_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS, 5)}},
    }

    # HBM-side traffic of the same kernel from the committed rocprofv3 --pmc passes
    # (cannot be collected live: PMC runs serialise kernels); see profiles/.
    tpath = os.path.join(REPO, "profiles", f"traffic_{args.workload}.json")
    if world == 1 and args.variant == 0 and os.path.exists(tpath):
        prof = json.load(open(tpath))
        out["roofline"]["traffic"] = prof["traffic_bytes_per_launch"]
        if "valu_wave_instructions_per_launch" in prof:
            # what actually bounds the kernel: VALU issue.  Instructions per launch from the committed PMC pass,
            # issue rate from the micro-benchmark (1.0 ns per wave64 instruction per SIMD, 1024 SIMDs), over the
            # time one frame takes in the pipeline
            busy_ms = prof["valu_wave_instructions_per_launch"] * VALU_NS_PER_WAVE_INSTR * 1e-6 / N_SIMDS
            out["roofline"]["valu"]["issue_frac"] = round(busy_ms / job_ms, 4)
            out["roofline"]["valu"]["wave_instructions_per_launch"] = prof["valu_wave_instructions_per_launch"]

    if world == 1 and not args.no_cpu_baseline:
        # oracle A (the CPU restatement) on the same frame, bounded: every 2nd pixel row
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from oracles import OracleA
        A = OracleA()
        cores = min(16, os.cpu_count() or 1)
        dirs = d_dir.view(H, W, 3)[::2].contiguous().cpu().numpy().reshape(-1, 3)
        orgs = np.broadcast_to(np.asarray(camera.position, np.float32), dirs.shape).copy()
        reps = 12
        tc0 = time.perf_counter()
        for _ in range(reps):
            r = A.render(sc, dirs, orgs, has_tc, has_n, tex[0], tex[1], nthreads=cores)
        cpu_s = time.perf_counter() - tc0
        cpu_rays = reps * (len(dirs) + int((r["hit_id"] != 0xFFFFFFFF).sum()))
        out["cpu_baseline"] = {"value": round(cpu_rays / cpu_s / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                               "sample": f"oracle/oracle_a.c, {reps} passes over every 2nd pixel row of the same frame "
                                         f"({len(dirs)} pixels per pass, {cpu_rays} rays in all, {cpu_s:.1f} s wall on {cores} threads)"}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
