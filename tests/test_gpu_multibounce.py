"""BASELINE config 5's workload under test: the multi-sample, multi-bounce chain of raytracer_amd/multibounce.py
(jittered primaries -> trace -> (surface -> dispatcher -> shadow + continuation traces) x depth; the pieces restate
ray_dispatcher.frag:26-41, ray_trace.frag:131-143 and the frame loop of src/main.rs:156-203) on the bench scene.

  * 240x135, 4 samples (even and odd continuation kinds, non-centre sample offsets), depth 3: EVERY buffer of
    the chain equals the CPU oracle's bit for bit, the oracle running the whole chain on its own;
  * 3840x2160 (the config's frame size), 1 sample, depth 3: size-independent properties — both kernel variants
    produce identical buffers, pixels without a surface keep NO_RAY / the cleared hit record through every
    stage — and stage by stage against the oracle (ray generation and the dispatcher on the full frame, the
    per-ray stages on sampled rows)."""
import math

import numpy as np
import pytest

from common import bits
from raytracer_amd import host, scenes
from raytracer_amd.multibounce import MultiBounce, noise_offset_scale, sample_offset

pytestmark = pytest.mark.gpu

LIGHT = (0.0, 12.0, 0.0)
DEPTH = 3
MISS = np.float32(1e30)


@pytest.fixture(scope="module")
def atrium_scene():
    tri, pos, uv, nrm, cam = scenes.atrium()
    m = host.Model.from_arrays(tri, pos, uv, nrm)
    m.build_bvh(8)
    rng = np.random.default_rng(3)
    v = rng.normal(size=(512, 512, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    noise = np.clip(np.round((v * 0.5 + 0.5) * 255), 0, 255).astype(np.uint8)
    return m.scene_arrays(), cam, noise


def _camera(cam, w, h):
    d = np.asarray(cam["direction"], np.float32)
    d /= np.linalg.norm(d)
    return host.Camera(w / h, fov=math.radians(cam["fov_deg"]), position=cam["position"], direction=d)


def _oracle_chain(A, sc, noise, w, h, inv, cam_pos, s, depth_max):
    """The whole chain on the CPU oracle alone: dict (depth, stage) -> array, as MultiBounce.sample records it."""
    n = w * h
    out = {}
    cam = np.asarray(cam_pos, np.float32)
    d = A.raycreate(w, h, inv, host.NEAR, host.FAR, offset=sample_offset(s)).reshape(n, 3)
    org = np.tile(cam, (n, 1))
    hits = A.trace(sc, org, d, nthreads=16)["hits"]
    out[(0, "dir")], out[(0, "hits")] = d, hits
    for depth in range(1, depth_max + 1):
        pos, nm = A.surface_from_hits(sc, org, d, hits, 1)
        r = A.dispatch(pos.reshape(h, w, 3), nm.reshape(h, w, 4), noise, np.float32(LIGHT), cam, np.float32(noise_offset_scale(s, depth, w, h)))
        r = {k: v.reshape(n, 3) for k, v in r.items()}
        out[(depth, "position")], out[(depth, "normal_mat")] = pos, nm
        for k in ("org", "shadow_dir", "reflect_dir", "ambient_dir"):
            out[(depth, k)] = r[k]
        out[(depth, "shadow_hits")] = A.trace(sc, r["org"], r["shadow_dir"], nthreads=16)["hits"]
        if depth < depth_max:
            cont = r["reflect_dir"] if s % 2 == 0 else r["ambient_dir"]
            hits = A.trace(sc, r["org"], cont, nthreads=16)["hits"]
            out[(depth, "cont_hits")] = hits
            org, d = r["org"], cont
    return out


def test_multibounce_chain_equals_oracle_buffer_for_buffer(tracer, oracle_a, atrium_scene):
    sc, cam, noise = atrium_scene
    w, h = 240, 135
    camera = _camera(cam, w, h)
    inv = np.asarray(camera.inv_proj_view(), np.float32).reshape(16)
    tracer.upload_scene(sc)
    tracer.upload_noise_texture(noise)
    mb = MultiBounce(tracer, w, h, inv, camera.position, LIGHT, True, depth=DEPTH, lanes=2)
    try:
        for s in range(4):
            rec = {}
            mb.sample(s, record=rec)
            mb.synchronize()
            want = _oracle_chain(oracle_a, sc, noise, w, h, inv, camera.position, s, DEPTH)
            assert set(rec) == set(want)
            for key in sorted(want):
                assert np.array_equal(bits(rec[key].cpu().numpy()), bits(want[key])), (s, key)
            # the chain does something at every depth: surfaces, occluded shadow rays, continuation hits
            for depth in range(1, DEPTH + 1):
                surface = want[(depth, "normal_mat")][:, 3] != MISS
                assert surface.sum() > (0.5 if depth == 1 else 0.05) * w * h, (s, depth)
                assert (want[(depth, "shadow_hits")][surface, 0] < MISS).sum() > 100
        if s > 0:
            assert sample_offset(s) != (0.5, 0.5)
        assert mb.rays_traced() > 4 * w * h * 3
    finally:
        tracer.set_stream(0)


@pytest.mark.timeout(900)
def test_multibounce_full_size_4k_properties(tracer, oracle_a, atrium_scene):
    import torch
    sc, cam, noise = atrium_scene
    w, h = 3840, 2160
    n = w * h
    camera = _camera(cam, w, h)
    inv = np.asarray(camera.inv_proj_view(), np.float32).reshape(16)
    cam_pos = np.asarray(camera.position, np.float32)
    tracer.upload_scene(sc)
    tracer.upload_noise_texture(noise)
    s = 1                                                     # an odd sample: off-centre primaries, ambient continuation
    mb = MultiBounce(tracer, w, h, inv, cam_pos, LIGHT, True, depth=DEPTH, lanes=1)
    recs = []
    try:
        for variant in (0, 1):
            tracer.set_kernel_variant(variant)
            rec = {}
            mb.sample(s, record=rec)
            mb.synchronize()
            recs.append(rec)
    finally:
        tracer.set_kernel_variant(0)
        tracer.set_stream(0)
    a, b = recs
    # (1) both kernel variants: identical buffers, every stage
    assert set(a) == set(b) and len(a) == 2 + DEPTH * 7 + (DEPTH - 1)
    for key in a:
        assert torch.equal(a[key].view(torch.int32), b[key].view(torch.int32)), key
    del b, recs
    # (2) pixels without a surface stay NO_RAY / cleared through every later stage; surfaces thin out with depth
    clear = torch.tensor([1e30, 0.0, 0.0, 0.0], dtype=torch.float32, device=a[(0, "hits")].device)
    n_surface = []
    for depth in range(1, DEPTH + 1):
        no_surface = a[(depth, "normal_mat")][:, 3] == 1e30
        prev_hits = a[(0, "hits")] if depth == 1 else a[(depth - 1, "cont_hits")]
        assert torch.equal(no_surface, prev_hits[:, 0] == 1e30)
        for k in ("org", "shadow_dir", "reflect_dir", "ambient_dir", "position"):
            assert bool((a[(depth, k)][no_surface] == 0).all()), (depth, k)
        assert bool((a[(depth, "shadow_hits")][no_surface] == clear).all())
        if depth < DEPTH:
            assert bool((a[(depth, "cont_hits")][no_surface] == clear).all())
        n_surface.append(int((~no_surface).sum()))
    assert n > n_surface[0] > n_surface[1] > n_surface[2] > 0
    # (3) stage by stage against the oracle: full frame for ray generation and the dispatcher, sampled rows for the per-ray stages
    rows = np.arange(5, h, 97)
    idx = (rows[:, None] * w + np.arange(w)[None, :]).reshape(-1)
    t_idx = torch.from_numpy(idx).to(a[(0, "dir")].device)
    sub = lambda key: a[key][t_idx].cpu().numpy()      # noqa: E731
    full = lambda key: a[key].cpu().numpy()            # noqa: E731
    d_full = full((0, "dir"))
    assert np.array_equal(bits(d_full), bits(oracle_a.raycreate(w, h, inv, host.NEAR, host.FAR, offset=sample_offset(s)).reshape(n, 3)))
    org, d = np.tile(cam_pos, (len(idx), 1)), d_full[idx]
    del d_full
    hits = sub((0, "hits"))
    assert np.array_equal(bits(hits), bits(oracle_a.trace(sc, org, d, nthreads=16)["hits"]))
    for depth in range(1, DEPTH + 1):
        pos, nm = oracle_a.surface_from_hits(sc, org, d, hits, 1)
        assert np.array_equal(bits(pos), bits(sub((depth, "position")))) and np.array_equal(bits(nm), bits(sub((depth, "normal_mat")))), depth
        r = oracle_a.dispatch(full((depth, "position")).reshape(h, w, 3), full((depth, "normal_mat")).reshape(h, w, 4), noise, np.float32(LIGHT), cam_pos,
                              np.float32(noise_offset_scale(s, depth, w, h)))
        for k in ("org", "shadow_dir", "reflect_dir", "ambient_dir"):
            assert np.array_equal(bits(r[k].reshape(n, 3)), bits(full((depth, k)))), (depth, k)
        org = r["org"].reshape(n, 3)[idx]
        assert np.array_equal(bits(sub((depth, "shadow_hits"))), bits(oracle_a.trace(sc, org, r["shadow_dir"].reshape(n, 3)[idx], nthreads=16)["hits"])), depth
        if depth < DEPTH:
            d = r["ambient_dir"].reshape(n, 3)[idx]
            hits = sub((depth, "cont_hits"))
            assert np.array_equal(bits(hits), bits(oracle_a.trace(sc, org, d, nthreads=16)["hits"])), depth


def _shard_index(w, h, rank, world):
    """For every entry of rank's compact tile-major buffer: the scan-line index of its pixel, or -1 outside the frame."""
    tiles_x, tiles_y = (w + 31) // 32, (h + 31) // 32
    local = [g for g in range(tiles_x * tiles_y) if g % world == rank]
    idx = np.full((len(local), 32, 32), -1, np.int64)
    for j, g in enumerate(local):
        tx, ty = g % tiles_x, g // tiles_x
        ys, xs = np.mgrid[0:32, 0:32]
        x, y = tx * 32 + xs, ty * 32 + ys
        inside = (x < w) & (y < h)
        idx[j][inside] = (y * w + x)[inside]
    return idx.reshape(-1)


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharded_chain_equals_the_whole_frame_chain(tracer, atrium_scene, world):
    """BASELINE config 5 split over GPUs: every rank traces its interleaved 32 x 32 tiles of the frame through the same chain
    (rtamd_generate_rays_tiles_device, the dispatcher's tile mode, ray lists for everything else).  Every buffer of every rank's
    chain equals the whole-frame chain's — which test_multibounce_chain_equals_oracle_buffer_for_buffer pins to the oracle —
    at the pixels of its tiles, bit for bit; entries of edge tiles outside the 240 x 135 frame are NO_RAY / cleared."""
    sc, cam, noise = atrium_scene
    w, h = 240, 135
    camera = _camera(cam, w, h)
    inv = np.asarray(camera.inv_proj_view(), np.float32).reshape(16)
    tracer.upload_scene(sc)
    tracer.upload_noise_texture(noise)
    clear = {"hits": np.float32([1e30, 0, 0, 0]), "shadow_hits": np.float32([1e30, 0, 0, 0]), "cont_hits": np.float32([1e30, 0, 0, 0]),
             "normal_mat": np.float32([0, 0, 0, 1e30])}
    try:
        whole = MultiBounce(tracer, w, h, inv, camera.position, LIGHT, True, depth=DEPTH, lanes=1)
        total_rays = 0
        for s in (1, 2):                                           # an odd and an even sample: both continuation kinds, off-centre offsets
            full = {}
            whole.sample(s, record=full)
            whole.synchronize()
            full = {k: v.cpu().numpy() for k, v in full.items()}
            seen = np.zeros(w * h, np.int32)
            for rank in range(world):
                mb = MultiBounce(tracer, w, h, inv, camera.position, LIGHT, True, depth=DEPTH, lanes=1, tile_rank=rank, tile_world=world)
                rec = {}
                mb.sample(s, record=rec)
                mb.synchronize()
                idx = _shard_index(w, h, rank, world)
                inside = idx >= 0
                assert mb.n == len(idx) and mb.n_primary == int(inside.sum())
                seen[idx[inside]] += 1
                assert set(rec) == set(full)
                for key in sorted(full):
                    got = rec[key].cpu().numpy()
                    assert np.array_equal(bits(got[inside]), bits(full[key][idx[inside]])), (s, rank, key)
                    want_out = clear.get(key[1], np.zeros(got.shape[1], np.float32))
                    assert np.array_equal(bits(got[~inside]), bits(np.broadcast_to(want_out, got[~inside].shape))), (s, rank, key, "outside the frame")
                total_rays += mb.rays_traced()
            assert np.all(seen == 1)                                # the ranks' tiles partition the frame
        assert total_rays == whole.rays_traced()                   # the shards trace the whole frame's rays, no more
    finally:
        tracer.set_stream(0)


def test_samples_in_lock_step_equal_samples_one_by_one(tracer, atrium_scene):
    """MultiBounce.sample_batch: five samples whose traversals go out stage by stage as ONE rtamd_trace_batch_device launch (5
    primaries; 10 shadow + continuation lists per bounce) leave, sample by sample, the buffers sample() leaves — also on a tile shard."""
    import torch
    sc, cam, noise = atrium_scene
    w, h = 240, 135
    camera = _camera(cam, w, h)
    inv = np.asarray(camera.inv_proj_view(), np.float32).reshape(16)
    tracer.upload_scene(sc)
    tracer.upload_noise_texture(noise)
    samples = [0, 1, 2, 5, 8]
    try:
        for rank, world in ((0, 1), (1, 3)):
            mb = MultiBounce(tracer, w, h, inv, camera.position, LIGHT, True, depth=DEPTH, lanes=5, tile_rank=rank, tile_world=world)
            recs = [{} for _ in samples]
            mb.sample_batch(samples, records=recs)
            mb.synchronize()
            batch_rays = mb.rays_traced(reset=True)
            for k, s in enumerate(samples):
                one = {}
                mb.sample(s, lane=mb.lanes[0], record=one)
                mb.synchronize()
                assert set(one) == set(recs[k]) and len(one) == 2 + DEPTH * 7 + (DEPTH - 1)
                for key in one:
                    assert torch.equal(one[key].view(torch.int32), recs[k][key].view(torch.int32)), (world, s, key)
            assert mb.rays_traced() == batch_rays
    finally:
        tracer.set_stream(0)


def test_chain_as_one_c_call_equals_oracle_and_stage_by_stage(tracer, oracle_a, atrium_scene):
    """rtamd_multibounce_sample_batch_device — the whole chain as ONE C entry point (what a non-Python host calls) — leaves in
    the lanes' buffers, for max_depth 1, 2 and 3, exactly what the CPU oracle's chain holds at that point (so every buffer of
    every depth is compared: a run of depth d ends with depth d's G-buffer, targets and shadow records and depth d-1's
    continuation records), counts the traversals it starts, and prepares a lane's eye array itself."""
    import ctypes
    import torch
    from raytracer_amd import _ffi
    sc, cam, noise = atrium_scene
    w, h = 240, 135
    n = w * h
    camera = _camera(cam, w, h)
    inv = np.asarray(camera.inv_proj_view(), np.float32).reshape(16)
    tracer.upload_scene(sc)
    tracer.upload_noise_texture(noise)
    samples = [0, 3, 6]
    try:
        want = {s: _oracle_chain(oracle_a, sc, noise, w, h, inv, camera.position, s, DEPTH) for s in samples}
        for depth in (1, 2, 3):
            mb = MultiBounce(tracer, w, h, inv, camera.position, LIGHT, True, depth=depth, lanes=len(samples))
            for ln in mb.lanes:          # the library fills the eye array itself (rtamd_multibounce_prepare_lane_device)
                ln.cam_org.fill_(-1.0)
            torch.cuda.synchronize()
            desc = mb.c_desc()
            assert tracer._lib.rtamd_multibounce_entries(ctypes.byref(desc)) == n == tracer._lib.rtamd_multibounce_primary_rays(ctypes.byref(desc))
            tracer.set_stream(mb.lanes[0].stream.cuda_stream)
            for ln in mb.lanes:
                c = ln.c_lane()
                tracer._check(tracer._lib.rtamd_multibounce_prepare_lane_device(tracer._h, ctypes.byref(desc), ctypes.byref(c)))
            mb.sample_batch(samples)
            mb.synchronize()
            rays = 0
            for k, s in enumerate(samples):
                ln, o = mb.lanes[k], want[s]
                t = ln.targets[depth & 1]
                got = {"position": ln.pos, "normal_mat": ln.nm, "org": t["org"], "shadow_dir": t["shadow"], "reflect_dir": t["reflect"],
                       "ambient_dir": t["ambient"], "shadow_hits": ln.shadow_hits}
                for key, buf in got.items():
                    # a deeper oracle chain's buffers of depth d do not depend on max_depth, except that the last depth has no continuation
                    assert np.array_equal(bits(buf.cpu().numpy()), bits(o[(depth, key)])), (depth, s, key)
                prev = o[(0, "hits")] if depth == 1 else o[(depth - 1, "cont_hits")]
                assert np.array_equal(bits(ln.hits.cpu().numpy()), bits(prev)), (depth, s, "hits")
                assert np.array_equal(bits(ln.dir.cpu().numpy()), bits(o[(0, "dir")])), (depth, s, "dir")
                rays += n
                for d in range(1, depth + 1):
                    surfaces = int((o[(d, "normal_mat")][:, 3] != MISS).sum())
                    rays += surfaces * (2 if d < depth else 1)
            assert mb.rays_traced() == rays
    finally:
        tracer.set_stream(0)
