"""Multi-sample, multi-bounce frames (BASELINE config 5) as a chain of C-ABI calls on device buffers.

The reference's v0 pipeline (src/main.rs:156-203) is one bounce: G-buffer -> ray_dispatcher.frag ->
three ray_trace.frag passes.  Here the same two shaders' restatements repeat, with
rtamd_surface_from_hits_device turning a bounce's hit records into the next G-buffer (the reference
fills it with its rasteriser, src/main.rs:127-140).  Per sample s of a pixel:

    jittered primary  rtamd_generate_rays_tiles_device  (R2 sequence; sample 0 = the pixel centre)
    closest hit       rtamd_trace_image_batch_device               (ray_trace.frag:131-143; 8x8-pixel work items)
    for depth 1..D:
        surface       rtamd_surface_from_hits_device               (position, normal | material)
        dispatcher    rtamd_dispatch_rays_device                   (ray_dispatcher.frag:26-41)
        shadow ray    ... (org, shadow_dir)                        -> shadow_hits
        continuation  ... (org, reflect_dir)  for even s           (mirror, :41)
                      ... (org, ambient_dir)  for odd s            (normalize(n + noise*2-1), :42)
                      (not after the last depth)

No colour comes out of this: the reference's consumer of these records (shader.frag:120-124) is a TODO
stub, so parity is pinned at ray level — every buffer of the chain against the CPU oracle
(tests/test_gpu_multibounce.py).  There is no compute here: every step is a call into librtamd.so.
"""
import ctypes

import numpy as np

from . import _ffi
from .host import FAR, NEAR

R2_G = 0.7548776662466927   # plastic-number constant of the R2 low-discrepancy sequence


def sample_offset(s):
    """Sample position inside the pixel for sample index s (s = 0: the pixel centre the reference rasterises)."""
    return (0.5 + s * R2_G) % 1.0, (0.5 + s * R2_G * R2_G) % 1.0


def noise_offset_scale(s, depth, width, height, noise_size=512):
    """ray_dispatcher's uniform 5 (src/main.rs:161-166): a per-(sample, depth) texel offset, frame / noise scale."""
    return ((7 * s + 3 * depth) % noise_size / noise_size, (11 * s + 5 * depth) % noise_size / noise_size,
            width / noise_size, height / noise_size)


STAGES = ("dir", "hits", "position", "normal_mat", "org", "shadow_dir", "reflect_dir", "ambient_dir", "shadow_hits", "cont_hits")


class Lane:
    """One sample in flight: a stream and the buffers of the chain (torch tensors on the tracer's GPU)."""

    def __init__(self, torch, dev, n, cam_pos):
        self.stream = torch.cuda.Stream(device=dev)
        f = lambda c: torch.empty((n, c), dtype=torch.float32, device=dev)   # noqa: E731
        self.dir, self.hits, self.pos, self.nm = f(3), f(4), f(3), f(4)
        # two sets of dispatcher targets, used alternately: a bounce's continuation rays (its origin and its reflect / ambient
        # directions) must survive the next bounce's dispatcher pass
        self.targets = [{"org": f(3), "shadow": f(3), "reflect": f(3), "ambient": f(3)} for _ in range(2)]
        self.shadow_hits = f(4)
        self.cam_org = torch.tensor(cam_pos, dtype=torch.float32, device=dev).repeat(n, 1).contiguous()
        self.rays = torch.zeros((), dtype=torch.int64, device=dev)
        # the buffers were made on torch's current stream; their first use is on this lane's (non-blocking) stream
        self.stream.wait_stream(torch.cuda.current_stream(dev))

    def c_lane(self):
        """This lane's buffers as the C entry point takes them (rtamd_multibounce_lane)."""
        ln = _ffi.MultiBounceLane()
        ln.cam_org, ln.dir, ln.hits, ln.position, ln.normal_mat = (t.data_ptr() for t in (self.cam_org, self.dir, self.hits, self.pos, self.nm))
        for k in range(2):
            ln.org[k], ln.shadow_dir[k] = self.targets[k]["org"].data_ptr(), self.targets[k]["shadow"].data_ptr()
            ln.reflect_dir[k], ln.ambient_dir[k] = self.targets[k]["reflect"].data_ptr(), self.targets[k]["ambient"].data_ptr()
        ln.shadow_hits = self.shadow_hits.data_ptr()
        return ln


class MultiBounce:
    def __init__(self, tracer, width, height, inv_proj_view, cam_pos, light_pos, has_normals, depth=3, lanes=4, device=0, noise_size=512,
                 tile_rank=0, tile_world=1):
        """tile_world > 1: this object traces one rank's interleaved 32 x 32 tiles of the frame (scene replicated, every sample
        of a pixel on the GPU that owns its tile, no exchange between ranks); every buffer then holds local tiles * 1024
        entries in the compact tile-major layout of include/rtamd.h, bit for bit the whole-frame chain's values of those pixels."""
        import torch
        from .renderer import TILE_PIXELS
        self.torch = torch
        self.tr, self.W, self.H = tracer, int(width), int(height)
        self.rank, self.world = int(tile_rank), int(tile_world)
        self.n = self.W * self.H if self.world == 1 else tracer.local_tile_count(self.W, self.H, self.rank, self.world) * TILE_PIXELS
        self.inv = np.asarray(inv_proj_view, np.float32).reshape(16)
        self.cam = tuple(float(x) for x in cam_pos)
        self.light = tuple(float(x) for x in light_pos)
        self.has_n, self.depth, self.noise_size = int(bool(has_normals)), int(depth), int(noise_size)
        dev = torch.device("cuda", device)
        self.lanes = [Lane(torch, dev, self.n, self.cam) for _ in range(max(1, min(16, lanes)))]
        # primaries actually traced: the shard's pixels inside the frame (entries of edge tiles outside it are NO_RAY)
        tiles_x = (self.W + 31) // 32
        g = np.arange(self.n // TILE_PIXELS if self.world > 1 else 0, dtype=np.int64) * self.world + self.rank
        tw = np.minimum(32, self.W - (g % tiles_x) * 32)
        th = np.minimum(32, self.H - (g // tiles_x) * 32)
        self.n_primary = int((tw * th).sum()) if self.world > 1 else self.n

    def sample(self, s, lane=None, record=None):
        """Enqueue sample s on a lane's stream.  record: a dict that receives a device clone of every stage buffer,
        keyed (depth, stage) with depth 0 = the primary stage (for tests)."""
        torch, tr, n, W, H = self.torch, self.tr, self.n, self.W, self.H
        lane = lane if lane is not None else self.lanes[s % len(self.lanes)]

        def rec(depth, **bufs):
            if record is not None:
                for k, v in bufs.items():
                    record[(depth, k)] = v.clone()      # on the lane's stream, behind the call that filled v

        with torch.cuda.stream(lane.stream):
            tr.set_stream(lane.stream.cuda_stream)
            ox, oy = sample_offset(s)
            tr.generate_rays_tiles_device(W, H, self.inv, ox, oy, self.rank, self.world, lane.dir.data_ptr())
            tr.trace_image_batch_device([(n, lane.cam_org.data_ptr(), lane.dir.data_ptr(), lane.hits.data_ptr())], W, self.world)
            lane.rays += self.n_primary
            rec(0, dir=lane.dir, hits=lane.hits)
            org, d = lane.cam_org, lane.dir
            for depth in range(1, self.depth + 1):
                tr.surface_from_hits_device(n, org.data_ptr(), d.data_ptr(), lane.hits.data_ptr(), self.has_n, lane.pos.data_ptr(), lane.nm.data_ptr())
                nos = noise_offset_scale(s, depth, W, H, self.noise_size)
                t = lane.targets[depth & 1]
                tr.dispatch_rays_device(W, H, lane.pos.data_ptr(), lane.nm.data_ptr(), self.light, self.cam, nos, org_ptr=t["org"].data_ptr(),
                                        shadow_dir_ptr=t["shadow"].data_ptr(), reflect_dir_ptr=t["reflect"].data_ptr(), ambient_dir_ptr=t["ambient"].data_ptr(),
                                        tile_rank=self.rank, tile_world=self.world)
                surfaces = (lane.nm[:, 3] != 1e30).sum()
                tr.trace_image_batch_device([(n, t["org"].data_ptr(), t["shadow"].data_ptr(), lane.shadow_hits.data_ptr())], W, self.world)
                lane.rays += surfaces
                rec(depth, position=lane.pos, normal_mat=lane.nm, org=t["org"], shadow_dir=t["shadow"], reflect_dir=t["reflect"],
                    ambient_dir=t["ambient"], shadow_hits=lane.shadow_hits)
                if depth < self.depth:
                    cont = t["reflect"] if s % 2 == 0 else t["ambient"]
                    # hits is read by nothing else from here on: the continuation's records replace the previous bounce's
                    tr.trace_image_batch_device([(n, t["org"].data_ptr(), cont.data_ptr(), lane.hits.data_ptr())], W, self.world)
                    lane.rays += surfaces
                    rec(depth, cont_hits=lane.hits)
                    org, d = t["org"], cont
        return lane

    def c_desc(self):
        d = _ffi.MultiBounceDesc()
        d.width, d.height = self.W, self.H
        for i in range(16):
            d.inv_proj_view[i] = float(self.inv[i])
        d.near_plane, d.far_plane = NEAR, FAR
        for i in range(3):
            d.cam_pos[i], d.light_pos[i] = self.cam[i], self.light[i]
        d.has_normals, d.max_depth, d.noise_size = self.has_n, self.depth, self.noise_size
        d.tile_rank, d.tile_world = self.rank, self.world
        return d

    def sample_batch(self, samples, lanes=None, records=None):
        """Several samples in lock step: sample samples[k] uses the buffers of lanes[k], everything is enqueued on lanes[0]'s
        stream, and the traversals of one stage — all samples' primaries, then per bounce all shadow and continuation rays —
        go out as ONE rtamd_trace_image_batch_device launch (up to 8 samples: 16 ray lists), so that the slow tail of a list runs
        beside the other lists' rays instead of beside an idle GPU.  Same buffers, bit for bit, as sample() gives each sample.
        This is ONE call into the library (rtamd_multibounce_sample_batch_device, the chain as a C entry point).
        records: a list of dicts (one per sample), filled like sample()'s record — intermediate buffers can only be copied
        between the stages, so with records the same chain is issued stage by stage from here instead."""
        torch, tr, n, W, H = self.torch, self.tr, self.n, self.W, self.H
        lanes = list(lanes if lanes is not None else self.lanes[:len(samples)])
        if not 1 <= len(samples) <= 8 or len(lanes) != len(samples):
            raise ValueError("1..8 samples, one lane each")
        K = len(samples)
        if records is None:
            head = lanes[0]
            with torch.cuda.stream(head.stream):
                tr.set_stream(head.stream.cuda_stream)
                desc = self.c_desc()
                c_lanes = (_ffi.MultiBounceLane * K)(*[ln.c_lane() for ln in lanes])
                c_samples = (ctypes.c_uint32 * K)(*[int(s) for s in samples])
                tr._check(tr._lib.rtamd_multibounce_sample_batch_device(tr._h, ctypes.byref(desc), c_samples, K, c_lanes, ctypes.c_void_p(head.rays.data_ptr())))
            return head

        def rec(k, depth, **bufs):
            if records is not None:
                for key, v in bufs.items():
                    records[k][(depth, key)] = v.clone()

        head = lanes[0]
        with torch.cuda.stream(head.stream):
            tr.set_stream(head.stream.cuda_stream)
            for k, s in enumerate(samples):
                ox, oy = sample_offset(s)
                tr.generate_rays_tiles_device(W, H, self.inv, ox, oy, self.rank, self.world, lanes[k].dir.data_ptr())
            tr.trace_image_batch_device([(n, ln.cam_org.data_ptr(), ln.dir.data_ptr(), ln.hits.data_ptr()) for ln in lanes], W, self.world)
            head.rays += self.n_primary * K
            for k, ln in enumerate(lanes):
                rec(k, 0, dir=ln.dir, hits=ln.hits)
            cur = [(ln.cam_org, ln.dir) for ln in lanes]
            for depth in range(1, self.depth + 1):
                jobs, surfaces = [], 0
                for k, (s, ln) in enumerate(zip(samples, lanes)):
                    org, d = cur[k]
                    tr.surface_from_hits_device(n, org.data_ptr(), d.data_ptr(), ln.hits.data_ptr(), self.has_n, ln.pos.data_ptr(), ln.nm.data_ptr())
                    t = ln.targets[depth & 1]
                    tr.dispatch_rays_device(W, H, ln.pos.data_ptr(), ln.nm.data_ptr(), self.light, self.cam, noise_offset_scale(s, depth, W, H, self.noise_size),
                                            org_ptr=t["org"].data_ptr(), shadow_dir_ptr=t["shadow"].data_ptr(), reflect_dir_ptr=t["reflect"].data_ptr(),
                                            ambient_dir_ptr=t["ambient"].data_ptr(), tile_rank=self.rank, tile_world=self.world)
                    n_surf = (ln.nm[:, 3] != 1e30).sum()
                    surfaces = surfaces + n_surf * (2 if depth < self.depth else 1)
                    jobs.append((n, t["org"].data_ptr(), t["shadow"].data_ptr(), ln.shadow_hits.data_ptr()))
                    rec(k, depth, position=ln.pos, normal_mat=ln.nm)      # hits is overwritten by this stage's continuation trace
                for k, (s, ln) in enumerate(zip(samples, lanes)):
                    if depth < self.depth:
                        t = ln.targets[depth & 1]
                        cont = t["reflect"] if s % 2 == 0 else t["ambient"]
                        jobs.append((n, t["org"].data_ptr(), cont.data_ptr(), ln.hits.data_ptr()))
                        cur[k] = (t["org"], cont)
                tr.trace_image_batch_device(jobs, W, self.world)
                head.rays += surfaces
                for k, ln in enumerate(lanes):
                    t = ln.targets[depth & 1]
                    rec(k, depth, org=t["org"], shadow_dir=t["shadow"], reflect_dir=t["reflect"], ambient_dir=t["ambient"], shadow_hits=ln.shadow_hits)
                    if depth < self.depth:
                        rec(k, depth, cont_hits=ln.hits)
        return head

    def synchronize(self):
        for lane in self.lanes:
            lane.stream.synchronize()

    def rays_traced(self, reset=False):
        """Traversals actually started (NO_RAY entries are not counted) since the last reset."""
        self.synchronize()
        total = int(sum(int(lane.rays.item()) for lane in self.lanes))
        if reset:
            for lane in self.lanes:
                lane.rays.zero_()
        return total
