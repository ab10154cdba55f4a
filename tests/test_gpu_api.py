"""GPU tests of the C-ABI beyond fixture parity: error behaviour, tile sharding +
assembly, on-device ray generation, and full-size (BASELINE configs) checks through
size-independent properties: scheduling knobs and kernel variants never change a
bit of the output, shards re-assemble to the single-GPU frame, and sampled pixels
agree with the CPU oracle."""
import ctypes
import math

import numpy as np
import pytest

from common import bits, load_golden, rgb_err_255
from raytracer_amd import host, scenes
from raytracer_amd.renderer import HipRayTracer, RtamdError, assemble_tiles_host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def atrium():
    tri, pos, uv, nrm, cam = scenes.atrium()
    m = host.Model.from_arrays(tri, pos, uv, nrm)
    m.build_bvh()
    d = np.asarray(cam["direction"], np.float32)
    d /= np.linalg.norm(d)
    camera = host.Camera(16 / 9, fov=math.radians(cam["fov_deg"]), position=cam["position"], direction=d)
    return m.scene_arrays(), camera, (scenes.checker_texture(), scenes.noise_texture())


def test_errors():
    t = HipRayTracer(0)
    try:
        with pytest.raises(RtamdError) as e:
            t.trace(np.zeros((4, 3), np.float32), np.ones((4, 3), np.float32))
        assert e.value.code == -3                                   # RTAMD_ERR_NO_SCENE
        g = load_golden("cube")
        with pytest.raises(RtamdError) as e:
            t.upload_scene({"nodes": g["nodes"][:0], "triangles": g["triangles"], "positions": g["positions"]})
        assert e.value.code == -2                                   # RTAMD_ERR_EMPTY_SCENE
        bad = g["nodes"].copy()
        bad["a"][0] = 10_000
        with pytest.raises(RtamdError) as e:
            t.upload_scene({"nodes": bad, "triangles": g["triangles"], "positions": g["positions"]})
        assert e.value.code == -4 and "out of range" in str(e.value)  # RTAMD_ERR_MALFORMED_SCENE
        cyc = g["nodes"].copy()
        inner = np.where(cyc["is_leaf"] == 0)[0]
        cyc["b"][inner[-1]] = 0                                     # child points back at the root
        with pytest.raises(RtamdError) as e:
            t.upload_scene({"nodes": cyc, "triangles": g["triangles"], "positions": g["positions"]})
        assert e.value.code == -4
        tb = g["triangles"].copy()
        tb["p2"][3] = 99
        with pytest.raises(RtamdError) as e:
            t.upload_scene({"nodes": g["nodes"], "triangles": tb, "positions": g["positions"]})
        assert e.value.code == -4
        t.upload_scene(g["scene"])
        with pytest.raises(RtamdError) as e:
            t.render(8, 8, dir=np.ones((64, 3), np.float32), org=np.zeros((64, 3), np.float32), has_normals=True)
        assert e.value.code == -1                                   # scene has no normals
        with pytest.raises(RtamdError):
            t.upload_texture(16, np.zeros((2, 2, 3), np.uint8))         # slots are 0..15
        with pytest.raises(RtamdError):
            t.set_option("no_such_option", 1)
        assert t.trace(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)).shape == (0, 4)
    finally:
        t.close()


def test_library_first_then_torch_in_a_fresh_process():
    """Load order: a process that creates the tracer before it ever imports torch must still be able to use
    torch on the GPU (one HIP runtime per process, raytracer_amd/_ffi.py)."""
    import subprocess, sys, os
    code = ("import numpy as np\n"
            "from raytracer_amd.renderer import HipRayTracer\n"
            "t = HipRayTracer(0)\n"
            "import torch\n"
            "x = torch.arange(8, device='cuda:0', dtype=torch.float32)\n"
            "assert float(x.sum()) == 28.0\n"
            "t.close(); print('ok')\n")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=repo, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_launch_timing_option(tracer):
    g = load_golden("cube")
    tracer.upload_scene(g["scene"])
    org = np.zeros((64, 3), np.float32); d = np.tile(np.float32([0, 0, -1]), (64, 1))
    want = tracer.trace(org, d)
    assert tracer.last_kernel_ms() > 0
    try:
        tracer.set_option("launch_timing", 0)
        assert np.array_equal(bits(tracer.trace(org, d)), bits(want))
        with pytest.raises(RtamdError):
            tracer.last_kernel_ms()
    finally:
        tracer.set_option("launch_timing", 1)
    tracer.trace(org, d)
    assert tracer.last_kernel_ms() > 0


def test_scene_info_and_single_leaf_root(tracer):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    m = host.Model.from_arrays(np.array([[0, 1, 2, 0]], np.uint32), pos)
    m.build_bvh()
    tracer.upload_scene(m.scene_arrays())
    info = tracer.scene_info()
    assert (info.triangle_count, info.inner_node_count, info.max_depth) == (1, 0, 0)
    hits = tracer.trace(np.array([[0.2, 0.2, 1], [5, 5, 1]], np.float32), np.array([[0, 0, -1], [0, 0, -1]], np.float32))
    assert hits[0, 0] == 1.0 and bits(hits)[0, 3] == 0 and hits[1, 0] == np.float32(1e30)


def test_big_leaves_use_the_overflow_table(tracer, oracle_a):
    """20 identical-centroid triangles cannot be split (one-sided partition, bvh.rs:232):
    one leaf with count 20 > 15 exercises the BigLeaf path."""
    pos = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)
    k = 20
    allpos = np.concatenate([pos + np.float32([0, 0, 0.01 * i]) * 0 for i in range(k)])  # identical triangles
    tri = np.array([[3 * i, 3 * i + 1, 3 * i + 2, 0] for i in range(k)], np.uint32)
    m = host.Model.from_arrays(tri, allpos)
    m.build_bvh()
    sc = m.scene_arrays()
    assert sc["nodes"]["b"].max() == k
    tracer.upload_scene(sc)
    assert tracer.scene_info().big_leaf_count == 1
    rng = np.random.default_rng(0)
    org = np.concatenate([rng.uniform(-1, 1, (256, 2)), np.full((256, 1), 2.0)], 1).astype(np.float32)
    d = np.tile(np.float32([0, 0, -1]), (256, 1))
    assert np.array_equal(bits(tracer.trace(org, d)), bits(oracle_a.trace(sc, org, d)["hits"]))


def _chain_scene(depth):
    """Hand-made BVH in the reference's buffer format (types.rs:101-109): a chain of `depth` inner
    nodes, node k = {leaf of the plane triangle at z = k, rest of the chain}.  A ray travelling
    towards -z enters the rest first at every level and defers the leaf: `depth` stacked entries."""
    n_tri = depth + 1
    pos = np.concatenate([np.float32([[-3, -3, z], [3, -3, z], [0, 3, z]]) for z in range(n_tri)])
    tri = np.array([[3 * i, 3 * i + 1, 3 * i + 2, 0] for i in range(n_tri)], np.uint32).view(host.TRIANGLE_DTYPE).reshape(-1)
    nodes = np.zeros(2 * depth + 1, host.NODE_DTYPE)
    for k in range(depth):
        nodes[2 * k] = ((-3, -3, k, 3, 3, depth), 0, 2 * k + 1, 2 * k + 2)
        nodes[2 * k + 1] = ((-3, -3, k, 3, 3, k), 1, k, 1)
    nodes[2 * depth] = ((-3, -3, depth, 3, 3, depth), 1, depth, 1)
    return {"nodes": nodes, "triangles": tri, "positions": pos, "tex_coords": None, "normals": None}


@pytest.mark.parametrize("depth", [17, 40, 99])
def test_deep_traversal_stack_matches_oracle(tracer, oracle_a, depth):
    """Stacks deeper than the 16 entries held in LDS: the oldest entries move out to the HBM
    columns and come back on the way up (rt_kernels.hip StackGlobal).  Rays towards -z stack one
    entry per level, rays towards +z none, oblique rays something in between, so the lanes of a
    wave sit at different depths.  Both kernel variants, hit records bit-exact.  99 levels is the
    deepest chain the shader's 100-entry stack (ray_trace_and_shade.frag:12) holds: it pushes both
    children before it pops the nearer one."""
    sc = _chain_scene(depth)
    tracer.upload_scene(sc)
    assert tracer.scene_info().max_depth == depth
    rng = np.random.default_rng(depth)
    n = 4096
    xy = rng.uniform(-3.5, 3.5, (n, 2))
    down = np.concatenate([xy, np.full((n, 1), depth + 5.0)], 1), np.tile([0.0, 0.0, -1.0], (n, 1))
    up = np.concatenate([xy, np.full((n, 1), -5.0)], 1), np.tile([0.0, 0.0, 1.0], (n, 1))
    o3 = np.concatenate([rng.uniform(-4, 4, (n, 2)), rng.choice([-5.0, depth + 5.0], (n, 1))], 1)
    tgt = np.concatenate([rng.uniform(-3, 3, (n, 2)), rng.uniform(0, depth, (n, 1))], 1)
    d3 = tgt - o3
    d3 /= np.linalg.norm(d3, axis=1, keepdims=True)
    org = np.concatenate([down[0], up[0], o3]).astype(np.float32)
    d = np.concatenate([down[1], up[1], d3]).astype(np.float32)
    perm = rng.permutation(len(org))                      # mix the three kinds inside every wave
    org, d = org[perm], d[perm]
    want = oracle_a.trace(sc, org, d, nthreads=8)["hits"]
    assert (want[:, 0] < 1e30).mean() > 0.4
    try:
        for variant in (0, 1):
            tracer.set_kernel_variant(variant)
            assert np.array_equal(bits(tracer.trace(org, d)), bits(want)), variant
    finally:
        tracer.set_kernel_variant(0)
    a = oracle_a.render(sc, d, org, 0, 0, nthreads=8)
    r = tracer.render(len(org), 1, dir=d, org=org, has_tex_coords=0, has_normals=0)
    assert np.array_equal(a["hit_id"], r["hit_id"]) and np.array_equal(bits(a["hit_t"]), bits(r["hit_t"]))
    assert rgb_err_255(a["rgba"], r["rgba"]).max() <= 0.05


def test_deepest_accepted_tree_and_rejection_beyond(tracer):
    """128 levels (beyond the shader's own stack, so no oracle): 112 entries per lane pass through the
    HBM columns.  The two kernel variants keep their stacks differently (LDS ring + HBM columns /
    LDS + private scratch) and must agree bit for bit; rays straight down hit the top plane at t = 5."""
    depth = 128
    tracer.upload_scene(_chain_scene(depth))
    rng = np.random.default_rng(7)
    n = 8192
    org = np.concatenate([rng.uniform(-3.5, 3.5, (n, 2)), np.full((n, 1), depth + 5.0)], 1).astype(np.float32)
    d = np.tile(np.float32([0, 0, -1]), (n, 1))
    d[1::2] = np.float32([0.01, 0.02, -1]) / np.linalg.norm([0.01, 0.02, -1]).astype(np.float32)
    got = tracer.trace(org, d)
    try:
        tracer.set_kernel_variant(1)
        assert np.array_equal(bits(tracer.trace(org, d)), bits(got))
    finally:
        tracer.set_kernel_variant(0)
    inside = (np.abs(org[::2, 0]) < 1.0) & (np.abs(org[::2, 1]) < 0.9)
    assert inside.sum() > 100
    assert np.all(got[::2][inside, 0] == 5.0) and np.all(bits(got[::2])[inside, 3] == depth)
    with pytest.raises(RtamdError) as e:
        tracer.upload_scene(_chain_scene(129))
    assert e.value.code == -4 and "depth" in str(e.value)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_tile_shards_assemble_to_the_full_frame(tracer, world):
    g = load_golden("bunny")
    h, w = g["dir"].shape[:2]
    tracer.upload_scene(g["scene"])
    full = tracer.render(w, h, dir=g["dir"], org=g["org"], has_normals=1)
    parts = [tracer.render(w, h, dir=g["dir"], org=g["org"], has_normals=1, tile_rank=r, tile_world=world) for r in range(world)]
    per_rank = tracer.tiles_per_rank(w, h, world) * 1024
    pad = lambda a: np.concatenate([a, np.zeros((per_rank - len(a),) + a.shape[1:], a.dtype)])
    frame = assemble_tiles_host([pad(p["rgba"]) for p in parts], w, h, world, 4).reshape(-1, 4)
    assert np.array_equal(bits(frame), bits(full["rgba"]))
    ids = assemble_tiles_host([pad(p["hit_id"].reshape(-1, 1)) for p in parts], w, h, world, 1).reshape(-1)
    assert np.array_equal(ids, full["hit_id"])
    # device-side assembly (root side of the RCCL gather)
    import torch
    gathered = torch.from_numpy(np.concatenate([pad(p["rgba"]) for p in parts])).cuda()
    out = torch.zeros((w * h, 4), dtype=torch.float32, device="cuda")
    tracer.assemble_tiles_device(gathered.data_ptr(), w, h, world, 4, out.data_ptr())
    tracer.synchronize()
    assert np.array_equal(bits(out.cpu().numpy()), bits(full["rgba"]))
    # a batch of 3 frames moved by one gather: [rank][frame][pixels] -> [frame][scan lines]; frame f is the frame scaled by f + 1
    blocks = np.stack([np.stack([pad(p["rgba"]) * np.float32(f + 1) for f in range(3)]) for p in parts])
    gathered = torch.from_numpy(blocks).cuda()
    outs = torch.zeros((3, w * h, 4), dtype=torch.float32, device="cuda")
    tracer.assemble_tile_batch_device(gathered.data_ptr(), w, h, world, 4, 3, outs.data_ptr())
    tracer.synchronize()
    for f in range(3):
        assert np.array_equal(bits(outs[f].cpu().numpy()), bits(full["rgba"] * np.float32(f + 1)))


@pytest.mark.parametrize("w,h", [(72, 40), (70, 33), (1920, 1080), (4, 1)])
def test_rgba8_share_assembly_equals_the_host_assembly(tracer, w, h):
    """One word per pixel (the display format that crosses xGMI): widths that are multiples of 4 take the four-pixels-per-thread kernel,
    others and buffers that are not 16-byte aligned the one-pixel kernel; both equal the host restatement, single frames and batches."""
    import torch
    rng = np.random.default_rng(w * 1000 + h)
    for world in (1, 2, 3):
        per_rank = tracer.tiles_per_rank(w, h, world) * 1024
        n_frames = 3
        shares = rng.integers(0, 2**32, size=(world, n_frames, per_rank), dtype=np.uint32)
        expect = [assemble_tiles_host([shares[r, f].reshape(-1, 1) for r in range(world)], w, h, world, 1).reshape(-1) for f in range(n_frames)]
        for shift in (0, 1):       # 1: source and destination one word off a 16-byte boundary
            src = torch.zeros((shares.size + 4,), dtype=torch.int32, device="cuda")
            src[shift:shift + shares.size] = torch.from_numpy(shares.reshape(-1).view(np.int32)).cuda()
            dst = torch.full((n_frames * w * h + 4,), -1, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            tracer.assemble_tile_batch_device(src.data_ptr() + 4 * shift, w, h, world, 1, n_frames, dst.data_ptr() + 4 * shift)
            tracer.synchronize()
            got = dst.cpu().numpy().view(np.uint32)
            for f in range(n_frames):
                assert np.array_equal(got[shift + f * w * h: shift + (f + 1) * w * h], expect[f]), (world, shift, f)
            assert np.all(got[:shift] == 0xFFFFFFFF) and np.all(got[shift + n_frames * w * h:] == 0xFFFFFFFF)      # nothing outside the frames
        one = torch.from_numpy(np.ascontiguousarray(shares[:, 0]).reshape(-1).view(np.int32)).cuda()
        out = torch.zeros((w * h,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        tracer.assemble_tiles_device(one.data_ptr(), w, h, world, 1, out.data_ptr())
        tracer.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), expect[0]), world


def test_on_device_ray_generation_equals_host_and_oracle(tracer, oracle_a):
    cam = host.Camera(16 / 9, fov=math.radians(75), position=(0.5, 1.5, -2.0), direction=(0.2, -0.1, 0.97))
    inv = cam.inv_proj_view()
    d_host, o_host = cam.primary_rays(160, 90)
    d_dev = tracer.generate_rays(160, 90, inv)
    assert np.array_equal(bits(d_dev), bits(d_host))
    assert np.array_equal(bits(d_dev), bits(oracle_a.raycreate(160, 90, inv, host.NEAR, host.FAR)))
    g = load_golden("teapot")
    tracer.upload_scene(g["scene"])
    c2 = host.Camera(160 / 90, position=g["org"][0, 0], direction=g["dir"][60, 80])
    a = tracer.render(160, 90, camera=(c2.inv_proj_view(), c2.position))
    dd, oo = c2.primary_rays(160, 90)
    b = tracer.render(160, 90, dir=dd, org=oo)
    assert np.array_equal(bits(a["rgba"]), bits(b["rgba"])) and np.array_equal(a["hit_id"], b["hit_id"])
    assert (a["hit_id"] != 0xFFFFFFFF).sum() > 500


def test_full_size_atrium_properties(tracer, oracle_a, atrium):
    """BASELINE config 3 (Sponza-class, 1080p, primary + shadow): every scheduling
    variant produces the same bits; sampled pixel rows equal the CPU oracle."""
    sc, camera, tex = atrium
    W, H = 1920, 1080
    tracer.upload_scene(sc)
    tracer.upload_texture(0, tex[0]); tracer.upload_texture(1, tex[1])
    inv = camera.inv_proj_view()
    ref = None
    defaults = {"kernel_variant": 0, "dyn_thresh": 0, "leaf_thresh": 0, "num_queues": 8, "queue_stride": 64,
                "max_blocks_per_cu": 0, "shadow_anyhit": 1, "queue_block": 256}
    try:
        tracer.set_option("shadow_anyhit", 0)   # reference-length shadow traversals: counters comparable with the oracle
        full = tracer.render(W, H, camera=(inv, camera.position), has_tex_coords=1, has_normals=1, want=("rgba", "hit_id", "hit_t", "counts"))
        tracer.set_option("shadow_anyhit", 1)
        for opts in ({"kernel_variant": 1}, {"kernel_variant": 0, "dyn_thresh": 1, "leaf_thresh": 4, "num_queues": 8},
                     {"kernel_variant": 0, "dyn_thresh": 40, "leaf_thresh": 64, "num_queues": 1, "queue_stride": 16},
                     {"kernel_variant": 0, "dyn_thresh": 64, "leaf_thresh": 1, "num_queues": 3, "max_blocks_per_cu": 5, "queue_block": 7},
                     {"kernel_variant": 0, "dyn_thresh": 1, "leaf_thresh": 24},
                     {"kernel_variant": 0, "dyn_thresh": 1, "leaf_thresh": 8, "num_queues": 16, "max_blocks_per_cu": 0},
                     {"kernel_variant": 0, "dyn_thresh": 32, "leaf_thresh_coherent": 33, "leaf_thresh_refill": 3},
                     {"kernel_variant": 0, "dyn_thresh": 0, "leaf_thresh": 0}, {"kernel_variant": 0, "dyn_thresh": 0}):
            for k, v in opts.items():
                tracer.set_option(k, v)
            r = tracer.render(W, H, camera=(inv, camera.position), has_tex_coords=1, has_normals=1, want=("rgba", "hit_id", "hit_t", "counts"))
            if ref is None:
                ref = r
            else:
                for key in ("rgba", "hit_id", "hit_t", "counts"):
                    assert np.array_equal(ref[key].view(np.uint32), r[key].view(np.uint32)), (opts, key)
    finally:
        for k, v in defaults.items():
            tracer.set_option(k, v)
    for key in ("rgba", "hit_id", "hit_t"):   # stopping shadow rays at the first occluder changes no output
        assert np.array_equal(ref[key].view(np.uint32), full[key].view(np.uint32)), key
    assert np.all(ref["counts"] <= full["counts"]) and ref["counts"].sum() < full["counts"].sum()
    ref = full
    hit = ref["hit_id"] != 0xFFFFFFFF
    assert 0.9 < hit.mean() < 0.99 and np.all(ref["rgba"][:, 3] == 1)
    dirs, orgs = camera.primary_rays(W, H)
    rows = [7, 300, 541, 700, 1079]
    sel = np.concatenate([np.arange(y * W, (y + 1) * W) for y in rows])
    a = oracle_a.render(sc, dirs.reshape(-1, 3)[sel], orgs.reshape(-1, 3)[sel], 1, 1, tex[0], tex[1], nthreads=8)
    assert np.array_equal(a["hit_id"], ref["hit_id"][sel]) and np.array_equal(bits(a["hit_t"]), bits(ref["hit_t"][sel]))
    assert np.array_equal(a["counts"], ref["counts"][sel])
    assert rgb_err_255(a["rgba"], ref["rgba"][sel]).max() <= 0.05


def test_full_size_bunny_class_1080p(tracer, oracle_a):
    """BASELINE config 2: bunny subdivided twice (80 016 triangles), 1080p, white albedo."""
    g = load_golden("bunny")
    t = g["triangles"]
    tri, pos, nrm = scenes.subdivide(np.stack([t["p0"], t["p1"], t["p2"], t["mat_idx"]], 1), g["positions"], g["normals"], levels=2)
    m = host.Model.from_arrays(tri, pos, None, nrm)
    m.build_bvh()
    sc = m.scene_arrays()
    tracer.upload_scene(sc)
    white = np.full((1, 1, 3), 255, np.uint8)
    tracer.upload_texture(0, white); tracer.upload_texture(1, white)
    lo, hi = pos.min(0), pos.max(0)
    c = (lo + hi) / 2
    eye = c + 0.9 * np.linalg.norm(hi - lo) / 2 * np.float32([0.7, 0.3, 0.65])
    d = (c - eye) / np.linalg.norm(c - eye)
    cam = host.Camera(16 / 9, fov=math.radians(60), position=eye, direction=d)
    W, H = 1920, 1080
    r = tracer.render(W, H, camera=(cam.inv_proj_view(), cam.position), has_normals=1)
    assert (r["hit_id"] != 0xFFFFFFFF).mean() > 0.2
    dirs, orgs = cam.primary_rays(W, H)
    sel = np.concatenate([np.arange(y * W, (y + 1) * W) for y in (200, 540, 541, 800)])
    a = oracle_a.render(sc, dirs.reshape(-1, 3)[sel], orgs.reshape(-1, 3)[sel], 0, 1, nthreads=8)
    assert np.array_equal(a["hit_id"], r["hit_id"][sel]) and np.array_equal(bits(a["hit_t"]), bits(r["hit_t"][sel]))
    assert rgb_err_255(a["rgba"], r["rgba"][sel]).max() <= 0.05


def test_soup_incoherent_trace_matches_oracle(tracer, oracle_a):
    """BASELINE config 4 stand-in at test size (200 k of the 10 M triangles): incoherent BVH,
    no normals / uvs, random rays; bit-exact hit records."""
    tri, pos, _ = scenes.triangle_soup(200_000, seed=1, extent=30.0)
    m = host.Model.from_arrays(tri, pos)
    m.build_bvh()
    sc = m.scene_arrays()
    tracer.upload_scene(sc)
    rng = np.random.default_rng(2)
    n = 100_000
    org = rng.uniform(-35, 35, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    got = tracer.trace(org, d)
    want = oracle_a.trace(sc, org, d, nthreads=8)["hits"]
    assert (want[:, 0] < 1e30).mean() > 0.3
    assert np.array_equal(bits(got), bits(want))


def test_frames_in_flight_on_several_streams(tracer, atrium):
    """Launches issued on different streams overlap (frames in flight): each stream has its
    own launch scratch in the context (8 sets), a further stream takes over the least recently
    used one.  Every frame must equal the frame rendered alone on the context's own stream."""
    import torch
    sc, camera, tex = atrium
    W, H = 640, 360
    tracer.upload_scene(sc)
    tracer.upload_texture(0, tex[0]); tracer.upload_texture(1, tex[1])
    inv = camera.inv_proj_view()
    dev = torch.device("cuda", 0)
    n = W * H
    alone = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    tracer.set_stream(None)
    tracer.render_device(W, H, camera=(inv, camera.position), has_tex_coords=True, has_normals=True, rgba_ptr=alone.data_ptr())
    tracer.synchronize()
    assert float(alone[:, 3].min()) == 1.0
    streams = [torch.cuda.Stream(device=dev) for _ in range(10)]
    outs = [torch.zeros((n, 4), dtype=torch.float32, device=dev) for _ in range(10)]
    try:
        for rnd in range(3):
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()
            for k in range(20):
                tracer.set_stream(streams[k % 10].cuda_stream)
                tracer.render_device(W, H, camera=(inv, camera.position), has_tex_coords=True, has_normals=True, rgba_ptr=outs[k % 10].data_ptr())
            torch.cuda.synchronize()
            for o in outs:
                assert torch.equal(o.view(torch.int32), alone.view(torch.int32))
        tracer.set_stream(streams[1].cuda_stream)
        assert tracer.last_kernel_ms() > 0
    finally:
        tracer.set_stream(None)


def test_display_pass_is_the_gl_unorm8_conversion(tracer):
    """rtamd_display_rgba8_device against raytracer_amd.assets.frame_to_rgb8 (clamp, x255, round to nearest even):
    random colours, exact halves, values outside [0,1], infinities and NaN; ragged pixel counts."""
    import torch
    from raytracer_amd import assets
    rng = np.random.default_rng(4)
    for n in (1, 63, 4097):
        x = rng.uniform(-0.2, 1.2, (n, 4)).astype(np.float32)
        k = min(n, 12)
        x[:k, 0] = (np.arange(k, dtype=np.float32) + 0.5) / 255                     # exact ties after x255
        x[:k, 1] = [np.inf, -np.inf, np.nan, 0.0, -0.0, 1.0, 0.5, 1e-30, 2.0, -1.0, 0.999999, 0.0019607844][:k]
        d = torch.from_numpy(x).cuda()
        out = torch.zeros((n,), dtype=torch.int32, device="cuda")
        tracer.display_rgba8_device(d.data_ptr(), n, out.data_ptr())
        tracer.synchronize()
        got = out.cpu().numpy().view(np.uint8).reshape(n, 4)
        a = np.nan_to_num(x, nan=0.0, posinf=1.0, neginf=0.0)
        want = np.rint(np.clip(a, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
        assert np.array_equal(got, want)
        assert np.array_equal(got[:, :3], assets.frame_to_rgb8(x, n, 1).reshape(n, 3))


@pytest.mark.parametrize("variant", [0, 1])
def test_fused_rgba8_output_equals_the_display_pass(tracer, atrium, variant):
    """rtamd_frame_out.rgba8: the kernel's own 8-bit pixel equals rtamd_display_rgba8_device applied to its float
    pixel and the host writer's conversion; whole frame and a tile shard."""
    import torch
    from raytracer_amd import assets
    sc, camera, tex = atrium
    W, H = 333, 187
    tracer.upload_scene(sc)
    tracer.upload_texture(0, tex[0]); tracer.upload_texture(1, tex[1])
    inv = camera.inv_proj_view()
    try:
        tracer.set_kernel_variant(variant)
        r = tracer.render(W, H, camera=(inv, camera.position), has_tex_coords=1, has_normals=1, want=("rgba", "rgba8"))
        part = tracer.render(W, H, camera=(inv, camera.position), has_tex_coords=1, has_normals=1, tile_rank=1, tile_world=3, want=("rgba", "rgba8"))
    finally:
        tracer.set_kernel_variant(0)
    for res in (r, part):
        n = len(res["rgba"])
        d = torch.from_numpy(res["rgba"]).cuda()
        out = torch.zeros((n,), dtype=torch.int32, device="cuda")
        tracer.display_rgba8_device(d.data_ptr(), n, out.data_ptr())
        tracer.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), res["rgba8"])
    assert np.array_equal(r["rgba8"].view(np.uint8).reshape(H, W, 4)[::-1, :, :3], assets.frame_to_rgb8(r["rgba"], W, H))
    assert np.all(r["rgba8"] >> 24 == 255)
    with pytest.raises(RtamdError):
        tracer.render(W, H, camera=(inv, camera.position), has_tex_coords=1, has_normals=1, want=("rgba8",))


def test_two_contexts_render_concurrently(tracer, atrium):
    """Two contexts of one process on the same GPU (each with its own scene copy and launch scratch),
    frames issued alternately on streams of their own: every frame equals the one rendered alone."""
    import torch
    sc, camera, tex = atrium
    W, H = 480, 270
    inv = camera.inv_proj_view()
    dev = torch.device("cuda", 0)
    tracer.upload_scene(sc)
    tracer.upload_texture(0, tex[0]); tracer.upload_texture(1, tex[1])
    alone = tracer.render(W, H, camera=(inv, camera.position), has_tex_coords=1, has_normals=1)["rgba"]
    other = HipRayTracer(0)
    try:
        other.upload_scene(sc)
        other.upload_texture(0, tex[0]); other.upload_texture(1, tex[1])
        ctxs = (tracer, other)
        streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
        outs = [torch.zeros((W * H, 4), dtype=torch.float32, device=dev) for _ in range(4)]
        for k in range(24):
            c = ctxs[k % 2]
            c.set_stream(streams[k % 4].cuda_stream)
            c.render_device(W, H, camera=(inv, camera.position), has_tex_coords=True, has_normals=True, rgba_ptr=outs[k % 4].data_ptr())
        torch.cuda.synchronize()
        for o in outs:
            assert np.array_equal(bits(o.cpu().numpy()), bits(alone))
    finally:
        tracer.set_stream(None)
        other.close()


@pytest.mark.parametrize("name", ["f16_640x480", "bunny_1080p", "atrium_1080p"])
def test_full_size_hit_records_equal_llvmpipe(tracer, name):
    """BASELINE configs 1-3 at their stated sizes against the reference itself: the hit records of the whole frame (bench.py's
    workload, both kernel variants) hash, pixel row by pixel row, to what ray_trace.frag gave on llvmpipe (golden_fullsize.npz) --
    bit-exact parity on all 307 200 / 2 073 600 pixels, not on a sample of rows -- and the fused shader's colours agree with
    ray_trace_and_shade.frag's on llvmpipe in every 16 x 16 block."""
    import bench
    from common import check_fullsize_colour, check_fullsize_rows
    model, camera, w, h, tex = bench.build_workload(name)
    sc = model.scene_arrays()
    tracer.upload_scene(sc)
    for slot, t in enumerate(tex):
        if t is not None:
            tracer.upload_texture(slot, t)
    dirs, orgs = camera.primary_rays(w, h)
    for variant in (0, 1):
        tracer.set_kernel_variant(variant)
        tracer.set_option("texture_filter_8bit", 1)
        try:
            hits = tracer.trace(orgs.reshape(-1, 3), dirs.reshape(-1, 3))
            r = tracer.render(w, h, dir=dirs, org=orgs, has_tex_coords=int(model.has_tex_coords()), has_normals=int(model.has_normals()), want=("rgba",))
        finally:
            tracer.set_kernel_variant(0)
            tracer.set_option("texture_filter_8bit", 0)
        check_fullsize_rows(name, sc, hits, w, h)
        # the fused shader's colour of the same frame, with the reference executor's texture filter: the mean RGB of every
        # 16 x 16 pixel block within 0.03/255 of llvmpipe's (per pixel the GPU's pow is good to 0.05/255)
        check_fullsize_colour(name, r["rgba"], w, h, 0.03)


def test_split_pipeline_4k_equals_llvmpipe(tracer):
    """The split pipeline of src/main.rs:156-203 at config 5's frame size against the reference's shaders on llvmpipe
    (golden_split4k.npz, make_golden.py:split4k): the atrium at 3840 x 2160 -- primary hit records, the G-buffer made from them,
    ray_dispatcher.frag's four targets (option "texture_filter_8bit": the noise fetch lands a few 1e-4 texels off the centres, where
    float bilinear and llvmpipe's 8-bit weights differ in the last bits of ambientDir) and ray_trace.frag on the shadow, reflection
    and ambient rays: nine buffers of 8.3 M pixels, every group of 16 rows hashing to llvmpipe's."""
    import bench
    from common import check_group_digests
    g = load_golden("split4k")
    w, h = (int(x) for x in g["size"])
    model, camera, _, _, _ = bench.build_workload("atrium_1080p")
    sc = model.scene_arrays()
    tracer.upload_scene(sc)
    tracer.upload_noise_texture(load_golden("dispatch2")["noise"])
    dirs, orgs = camera.primary_rays(w, h)
    dirs, orgs = dirs.reshape(-1, 3), orgs.reshape(-1, 3)
    primary = tracer.trace(orgs, dirs)
    check_group_digests(g, "primary", primary.reshape(h, w, 4))
    pos, nm = tracer.surface_from_hits(orgs, dirs, primary, has_normals=True)
    check_group_digests(g, "position", pos.reshape(h, w, 3))
    check_group_digests(g, "normal_mat", nm.reshape(h, w, 4))
    tracer.set_option("texture_filter_8bit", 1)
    try:
        d = tracer.dispatch_rays(pos.reshape(h, w, 3), nm.reshape(h, w, 4), g["light_pos"], g["camera_pos"], g["noise_offset_scale"])
    finally:
        tracer.set_option("texture_filter_8bit", 0)
    for k in ("org", "shadow_dir", "reflect_dir", "ambient_dir"):
        check_group_digests(g, k, d[k])
    for k in ("shadow_dir", "reflect_dir", "ambient_dir"):
        hits = tracer.trace(d["org"].reshape(-1, 3), d[k].reshape(-1, 3))
        check_group_digests(g, "hits_" + k, hits.reshape(h, w, 4))
        assert int((hits[:, 0] < 1e30).sum()) == int(g["n_hits_" + k])


@pytest.mark.timeout(900)
def test_full_size_soup_10m_4k(tracer, oracle_a):
    """BASELINE config 4 at full size: 10 000 000-triangle soup (seed 1, no uvs / normals), 3840x2160,
    camera at (0, 0, -250) looking +z, fovy 60.  Size-independent properties: the 8 interleaved tile
    shards of the multi-GPU split re-assemble to the single-GPU frame bit for bit, both kernel variants
    agree, sampled pixel rows equal the CPU oracle (hit records bit-exact, colour <= 0.05/255), and the hit records of the whole
    frame equal ray_trace.frag's on llvmpipe (row digests)."""
    tri, pos, cam = scenes.triangle_soup(10_000_000, seed=1)
    m = host.Model.from_arrays(tri, pos)
    del tri, pos
    m.build_bvh(16)
    sc = m.scene_arrays()
    del m
    assert len(sc["triangles"]) == 10_000_000
    tracer.upload_scene(sc)
    white = np.full((1, 1, 3), 255, np.uint8)      # the tracer fixture is shared: earlier tests left their albedo textures
    tracer.upload_texture(0, white); tracer.upload_texture(1, white)
    W, H = 3840, 2160
    camera = host.Camera(W / H, fov=math.radians(cam["fov_deg"]), position=cam["position"], direction=cam["direction"])
    inv = camera.inv_proj_view()
    full = tracer.render(W, H, camera=(inv, camera.position), want=("rgba", "hit_id", "hit_t"))
    hit = full["hit_id"] != 0xFFFFFFFF
    assert 0.05 < hit.mean() < 1.0
    try:
        tracer.set_kernel_variant(1)
        static = tracer.render(W, H, camera=(inv, camera.position), want=("rgba", "hit_id", "hit_t"))
    finally:
        tracer.set_kernel_variant(0)
    for k in ("rgba", "hit_id", "hit_t"):
        assert np.array_equal(full[k].view(np.uint32), static[k].view(np.uint32)), k
    del static
    world = 8
    parts = [tracer.render(W, H, camera=(inv, camera.position), tile_rank=r, tile_world=world, want=("rgba",))["rgba"] for r in range(world)]
    per = tracer.tiles_per_rank(W, H, world) * 1024
    blocks = [np.concatenate([p, np.zeros((per - len(p), 4), np.float32)]) for p in parts]
    frame = assemble_tiles_host(blocks, W, H, world, 4).reshape(-1, 4)
    assert np.array_equal(frame.view(np.uint32), full["rgba"].view(np.uint32))
    dirs, orgs = camera.primary_rays(W, H)
    rows = [3, 1080, 2100]
    sel = np.concatenate([np.arange(y * W, (y + 1) * W) for y in rows])
    a = oracle_a.render(sc, dirs.reshape(-1, 3)[sel], orgs.reshape(-1, 3)[sel], 0, 0, nthreads=16)
    assert np.array_equal(a["hit_id"], full["hit_id"][sel]) and np.array_equal(bits(a["hit_t"]), bits(full["hit_t"][sel]))
    assert rgb_err_255(a["rgba"], full["rgba"][sel]).max() <= 0.05
    # and against the reference itself: every row of the 8.3 M hit records hashes to ray_trace.frag's on llvmpipe (golden_fullsize.npz)
    from common import check_fullsize_rows
    hits = tracer.trace(orgs.reshape(-1, 3), dirs.reshape(-1, 3))
    check_fullsize_rows("soup_10m_4k", sc, hits, W, H)
    from common import check_fullsize_colour
    check_fullsize_colour("soup_10m_4k", full["rgba"], W, H, 0.03)     # the fused shader's colour: 16 x 16 block means vs llvmpipe
    assert np.array_equal(bits(hits[:, 0])[hit], bits(full["hit_t"])[hit]) and np.array_equal(bits(hits[:, 3])[hit], full["hit_id"][hit])


def test_coherence_feedback_separates_camera_rays_from_soup(tracer, atrium):
    """dyn_thresh = auto: every launch picks the lane re-fill threshold from what the previous launch on the
    stream measured (1 -> 32 under 50 % active lanes per visit; 32 is kept for 64 launches, then one probes at 1).  Camera
    frames of the atrium stay at 1, a triangle soup moves to 32; whatever the threshold, outputs do not change."""
    sc, camera, tex = atrium
    W, H = 640, 360
    inv = camera.inv_proj_view()

    def frames(n, **kw):
        out, sig = [], []
        for _ in range(n):
            out.append(tracer.render(W, H, **kw)["rgba"])
            sig.append(tracer.last_launch_coherence())
        for (u, l, t), (_, _, t_next) in zip(sig, sig[1:]):
            assert t in (1, 32) and (t_next == (32 if l < 0.5 else 1) if t == 1 else t_next in (1, 32)), sig
        assert all(np.array_equal(bits(out[0]), bits(f)) for f in out[1:])
        return sig

    tracer.set_option("dyn_thresh", 0)
    tracer.upload_scene(sc)
    tracer.upload_texture(0, tex[0]); tracer.upload_texture(1, tex[1])
    sig = frames(4, camera=(inv, camera.position), has_tex_coords=1, has_normals=1)
    assert all(t == 1 for _, _, t in sig) and sig[-1][0] > 0.4 and sig[-1][1] > 0.5
    tri, pos, cam = scenes.triangle_soup(400_000, seed=5, extent=30.0)
    m = host.Model.from_arrays(tri, pos)
    m.build_bvh(4)
    tracer.upload_scene(m.scene_arrays())
    c2 = host.Camera(W / H, fov=math.radians(60), position=(0.0, 0.0, -75.0), direction=(0.0, 0.0, 1.0))
    sig = frames(70, camera=(c2.inv_proj_view(), c2.position))
    ts = [t for _, _, t in sig]
    assert sig[0][1] < 0.5 and ts[0] == 1 and all(t == 32 for t in ts[1:65]) and ts[65] == 1 and ts[66] == 32


def _batch_vs_single(tracer, w, h, frames_in, has_tc, has_n, tile_rank=0, tile_world=1, with8=False, counts=True):
    """frames_in: list of dicts (dir/org device tensors, or camera).  Renders every frame alone (rtamd_render_device) and all of
    them in one launch (rtamd_render_batch_device) and returns the two lists of output dicts."""
    import torch
    from raytracer_amd.renderer import TILE_PIXELS
    n_out = w * h if tile_world == 1 else tracer.local_tile_count(w, h, tile_rank, tile_world) * TILE_PIXELS
    dev = torch.device("cuda", 0)

    def outputs():
        o = {"rgba": torch.full((n_out, 4), -3.0, dtype=torch.float32, device=dev), "hit_id": torch.full((n_out,), 77, dtype=torch.int32, device=dev),
             "hit_t": torch.full((n_out,), -3.0, dtype=torch.float32, device=dev)}
        if counts:
            o["counts"] = torch.full((n_out, 2), 77, dtype=torch.int32, device=dev)
        if with8:
            o["rgba8"] = torch.full((n_out,), 77, dtype=torch.int32, device=dev)
        return o

    def args(fin, o):
        a = {k + "_ptr": v.data_ptr() for k, v in o.items()}
        if "camera" in fin:
            a["camera"] = fin["camera"]
        else:
            a["dir_ptr"], a["org_ptr"] = fin["dir"].data_ptr(), fin["org"].data_ptr()
        return a

    for fin in frames_in:      # the library cannot know a device buffer's size: a short ray array would be read out of bounds
        assert "camera" in fin or (fin["dir"].numel() == 3 * w * h and fin["org"].numel() == 3 * w * h)
    single, batch = [outputs() for _ in frames_in], [outputs() for _ in frames_in]
    torch.cuda.synchronize()    # the fills ran on torch's stream, the library launches on the context's own non-blocking stream
    for fin, o in zip(frames_in, single):
        tracer.render_device(w, h, has_tex_coords=has_tc, has_normals=has_n, tile_rank=tile_rank, tile_world=tile_world, **args(fin, o))
    tracer.render_batch_device(w, h, [args(fin, o) for fin, o in zip(frames_in, batch)], has_tex_coords=has_tc, has_normals=has_n,
                               tile_rank=tile_rank, tile_world=tile_world)
    tracer.synchronize()
    return single, batch


def test_batch_of_frames_equals_frame_by_frame(tracer):
    """rtamd_render_batch_device: several frames (different rays, on-device cameras, different origins) through ONE persistent
    launch give, frame by frame, the bits of rtamd_render_device — hit buffers, colours, test counters, the 8-bit output;
    also with lane re-fill forced on (lanes of one wave then belong to different frames), on a tile shard, for ragged
    frames and for a full batch of 16."""
    import torch
    g = load_golden("f16")
    h, w = g["dir"].shape[:2]
    tracer.upload_scene(g["scene"])
    tracer.upload_texture(0, g["tex0"]); tracer.upload_texture(1, g["tex1"])
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)).to(dev)   # noqa: E731
    eye, d, fov = scenes.framing_camera(g["positions"], 0.55)
    cams = [host.Camera(w / h, fov=fov, position=eye, direction=d), host.Camera(w / h, fov=math.radians(70), position=eye * np.float32(1.3), direction=d)]
    frames = [{"dir": t(g["dir"]), "org": t(g["org"])},
              {"dir": t(g["dir"][::-1]), "org": t(g["org"])},                                        # upside down
              {"camera": (cams[0].inv_proj_view(), cams[0].position)},                               # rays made on the device
              {"dir": t(g["dir"][:, ::-1]), "org": t(g["org"] + np.float32([0.3, -0.2, 0.1]))},      # mirrored, moved eye
              {"camera": (cams[1].inv_proj_view(), cams[1].position)}]

    def same(single, batch):
        for k, (a, b) in enumerate(zip(single, batch)):
            for name in a:
                assert torch.equal(a[name].view(torch.int32), b[name].view(torch.int32)), (k, name)

    single, batch = _batch_vs_single(tracer, w, h, frames, 1, 1)
    same(single, batch)
    ids = [o["hit_id"] for o in batch]
    assert all(1000 < int((i != -1).sum()) < w * h for i in ids) and not torch.equal(ids[0], ids[1]) and not torch.equal(ids[2], ids[4])
    assert torch.equal(batch[2]["hit_id"], batch[0]["hit_id"])       # frame 2's camera is the fixture's: same picture from device-made rays
    try:
        tracer.set_option("dyn_thresh", 32)                           # re-fill idle lanes mid-item: a wave's lanes span frames
        same(*_batch_vs_single(tracer, w, h, frames, 1, 1))
        same(single, _batch_vs_single(tracer, w, h, frames, 1, 1)[1])
    finally:
        tracer.set_option("dyn_thresh", 0)
    same(*_batch_vs_single(tracer, w, h, frames[:3], 1, 1, tile_rank=1, tile_world=3, with8=True))
    # the other instantiations of the fused kernel (one-frame launches and batches are different ones of each): the executor's 8-bit
    # texture filter, a material table, both
    try:
        for f8, table in ((1, None), (0, [((0.3, 0.3, 0.3), (0.9, 0.6, 0.4), (0.2, 0.2, 0.2), 12.0, 1), ((0.5, 0.5, 0.5), (0.8, 0.8, 0.8), (0.4, 0.4, 0.4), 30.0, 0)]), (1, [((0.1, 0.2, 0.3), (0.7, 0.7, 0.2), (0.9, 0.9, 0.9), 5.0, -1)] * 2)):
            tracer.set_option("texture_filter_8bit", f8)
            tracer.upload_materials(table)
            same(*_batch_vs_single(tracer, w, h, frames[:3], 1, 1))
            same(*_batch_vs_single(tracer, w, h, frames[1:4], 1, 1, counts=False))
    finally:
        tracer.set_option("texture_filter_8bit", 0)
        tracer.upload_materials(None)
    # ragged frames, the full 16 of a batch; frame outputs may be partly absent
    rw, rh = 37, 19
    sub = lambda a, k: t(np.ascontiguousarray(a[5 * k:5 * k + rh, 8 * k:8 * k + rw]))   # noqa: E731  (k <= 15: inside the 160x120 fixture)
    small = [{"dir": sub(g["dir"], k), "org": sub(g["org"], k)} for k in range(16)]
    same(*_batch_vs_single(tracer, rw, rh, small, 1, 1, counts=False))
    one = torch.zeros((w * h, 4), dtype=torch.float32, device=dev)
    tracer.render_batch_device(w, h, [{"dir_ptr": frames[0]["dir"].data_ptr(), "org_ptr": frames[0]["org"].data_ptr(), "rgba_ptr": one.data_ptr()}], 1, 1)
    tracer.synchronize()
    assert torch.equal(one, single[0]["rgba"])                        # a batch of one is rtamd_render_device
    f0 = {"dir_ptr": frames[0]["dir"].data_ptr(), "org_ptr": frames[0]["org"].data_ptr(), "rgba_ptr": one.data_ptr()}
    with pytest.raises(RtamdError):
        tracer.render_batch_device(w, h, [f0] * 17, 1, 1)
    with pytest.raises(RtamdError):
        tracer.render_batch_device(w, h, [], 1, 1)


def test_config1_f16_640x480_matches_oracle(tracer, oracle_a):
    """BASELINE config 1 as stated: the reference's default scene (src/main.rs:23,30: f16.obj, its two BMP textures) at
    640x480, bench.py's `f16_640x480` workload.  The 160x120 llvmpipe fixture is this camera at a quarter of the
    resolution (tests/test_host_layer.py); here the full frame equals oracle A: hit buffers bit for bit, colour far
    inside the 1/255 bound."""
    import bench
    model, camera, w, h, tex = bench.build_workload("f16_640x480")
    assert (w, h) == (640, 480)
    sc = model.scene_arrays()
    tracer.upload_scene(sc)
    tracer.upload_texture(0, tex[0]); tracer.upload_texture(1, tex[1])
    dirs, orgs = camera.primary_rays(w, h)
    for variant in (0, 1):
        tracer.set_kernel_variant(variant)
        try:
            r = tracer.render(w, h, dir=dirs, org=orgs, has_tex_coords=1, has_normals=1)
            d = tracer.render(w, h, camera=(camera.inv_proj_view(), camera.position), has_tex_coords=1, has_normals=1)   # rays made on the device
        finally:
            tracer.set_kernel_variant(0)
        a = oracle_a.render(sc, dirs, orgs, 1, 1, tex[0], tex[1], nthreads=16)
        assert np.array_equal(r["hit_id"], a["hit_id"]) and np.array_equal(bits(r["hit_t"]), bits(a["hit_t"]))
        assert rgb_err_255(r["rgba"], a["rgba"]).max() <= 0.05
        for k in ("hit_id", "hit_t", "rgba"):
            assert np.array_equal(bits(r[k]) if k != "hit_id" else r[k], bits(d[k]) if k != "hit_id" else d[k]), k
    hit = r["hit_id"] != 0xFFFFFFFF
    assert 0.05 < hit.mean() < 0.9
    mats = sc["triangles"]["mat_idx"][r["hit_id"][hit]]
    assert (mats == 0).any() and (mats == 1).any()              # both textures are on screen


def test_device_buffers_are_range_checked(tracer):
    """A "_device" call whose buffer would run past the end of the device allocation it starts in is refused on the host
    (rtamd_set_option "validate_buffers", default on) instead of faulting on the GPU.  The buffers come straight from
    hipMalloc, one allocation each: a tensor of torch's caching allocator may sit inside a larger cached block, where
    running past the tensor is not running past the allocation (the check's documented limit)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def dev_alloc(nbytes, host=None):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0
        if host is None:
            assert hip.hipMemset(p, 0, ctypes.c_size_t(nbytes)) == 0
        else:
            assert host.nbytes == nbytes and hip.hipMemcpy(p, host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nbytes), 1) == 0
        return p.value

    g = load_golden("cube")
    tracer.upload_scene(g["scene"])
    w = h = 512
    host_rays = np.zeros((w * h, 3), np.float32)
    host_rays[:, 2] = -1.0
    rays = dev_alloc(w * h * 12, host_rays)
    out = dev_alloc(w * (h + 1) * 16)
    hits = dev_alloc(w * h * 16)
    try:
        tracer.render_device(w, h, rays, rays, rgba_ptr=out)           # fits
        tracer.synchronize()
        with pytest.raises(RtamdError) as e:                   # one more row of pixels than the ray buffers hold
            tracer.render_device(w, h + 1, rays, rays, rgba_ptr=out)
        assert e.value.code == -1 and "run past the end" in str(e.value)
        with pytest.raises(RtamdError):
            tracer.render_device_static(w, h + 1, rays, rays, rgba_ptr=out)
        with pytest.raises(RtamdError):
            tracer.render_batch_device(w, h + 1, [{"dir_ptr": rays, "org_ptr": rays, "rgba_ptr": out}] * 2)
        with pytest.raises(RtamdError):
            tracer.trace_device(w * h + 64, rays, rays, hits)
        with pytest.raises(RtamdError):
            tracer.trace_device(w * h, rays, rays, hits + 16 * 4096)     # output shifted past its end
        with pytest.raises(RtamdError):
            tracer.render_device(w, h, rays + 12 * w, rays, rgba_ptr=out)   # ray buffer entered one row late
        tracer.trace_device(w * h, rays, rays, hits)
        tracer.synchronize()
    finally:
        tracer.synchronize()
        for p in (rays, out, hits):
            assert hip.hipFree(ctypes.c_void_p(p)) == 0


@pytest.mark.parametrize("dyn_thresh", [0, 32])
def test_trace_batch_equals_list_by_list(tracer, dyn_thresh):
    """rtamd_trace_batch_device: 16 ray lists of ragged lengths (empty, shorter than a work item, not multiples of 64, NO_RAY
    entries) in one launch give, list by list, the bits of rtamd_trace_device; also with lane re-fill forced on, where the lanes
    of one wave belong to different lists.  The A/B baseline kernel takes the list-by-list path of the same entry point."""
    import torch
    g = load_golden("bunny")
    tracer.upload_scene(g["scene"])
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(17)
    lengths = [0, 1, 63, 64, 65, 1000, 4096, 5, 777, 12345, 0, 2, 129, 3000, 64 * 50, 191]
    lo, hi = g["scene"]["positions"].min(0), g["scene"]["positions"].max(0)
    jobs, bufs = [], []
    for k, n in enumerate(lengths):
        org = ((lo + hi) / 2 + rng.uniform(-1.0, 1.0, (n, 3)) * (hi - lo)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d[::7] = 0.0                                                     # NO_RAY
        t_org, t_dir = torch.from_numpy(org).to(dev), torch.from_numpy(d).to(dev)
        hits_b = torch.full((max(n, 1), 4), -3.0, dtype=torch.float32, device=dev)
        hits_s = torch.full((max(n, 1), 4), -5.0, dtype=torch.float32, device=dev)
        bufs.append((t_org, t_dir, hits_b, hits_s))
        jobs.append((n, t_org.data_ptr() if n else 0, t_dir.data_ptr() if n else 0, hits_b.data_ptr() if n else 0))
    tracer.set_option("dyn_thresh", dyn_thresh)
    try:
        for variant in (0, 1):
            tracer.set_kernel_variant(variant)
            for _, _, hb, _ in bufs:
                hb.fill_(-3.0)
            torch.cuda.synchronize()    # uploads and fills ran on torch's stream, the library launches on the context's own
            tracer.trace_batch_device(jobs)
            for n, (t_org, t_dir, _, hits_s) in zip(lengths, bufs):
                if n:
                    tracer.trace_device(n, t_org.data_ptr(), t_dir.data_ptr(), hits_s.data_ptr())
            tracer.synchronize()
            n_hit = 0
            for n, (_, _, hits_b, hits_s) in zip(lengths, bufs):
                if n:
                    assert torch.equal(hits_b.view(torch.int32), hits_s.view(torch.int32)), (variant, n)
                    n_hit += int((hits_b[:, 0] < 1e30).sum())
                else:
                    assert bool((hits_b == -3.0).all())
            assert n_hit > 1000
        with pytest.raises(RtamdError):
            tracer.trace_batch_device(jobs + [jobs[3]])                    # 17 lists
        tracer.trace_batch_device([jobs[0], jobs[10]])                     # nothing but empty lists: a no-op
        tracer.synchronize()
    finally:
        tracer.set_kernel_variant(0)
        tracer.set_option("dyn_thresh", 0)


@pytest.mark.parametrize("dyn_thresh", [0, 32])
def test_trace_image_batch_equals_plain_lists(tracer, dyn_thresh):
    """rtamd_trace_image_batch_device: ray lists that are images -- scan-line order with widths that are no multiple of 8, one
    row, one column, a single list, an empty one; and the tile-major shard layout -- are traced in 8x8-pixel work items and give,
    list by list, the bits of rtamd_trace_device in the same places; entries the items touch beyond a list's end stay untouched."""
    import torch
    g = load_golden("bunny")
    tracer.upload_scene(g["scene"])
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(23)
    lo, hi = g["scene"]["positions"].min(0), g["scene"]["positions"].max(0)

    def make(n):
        org = ((lo + hi) / 2 + rng.uniform(-1.0, 1.0, (n, 3)) * (hi - lo)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d[::11] = 0.0                                                    # NO_RAY
        t_org, t_dir = torch.from_numpy(org).to(dev), torch.from_numpy(d).to(dev)
        # 64 guard entries behind the list: an item that ran past the end would write there
        return t_org, t_dir, torch.full((n + 64, 4), -3.0, dtype=torch.float32, device=dev), torch.full((max(n, 1), 4), -5.0, dtype=torch.float32, device=dev)

    tracer.set_option("dyn_thresh", dyn_thresh)
    tracer.set_option("dyn_incoherent", 48 if dyn_thresh == 0 else 32)    # what auto switches to for these random rays
    try:
        cases = [("scan-line", 70, [70 * 37, 70 * 1, 0, 70 * 8, 70 * 9]), ("scan-line", 1, [1 * 130]), ("scan-line", 8, [8 * 8]),
                 ("scan-line", 640, [640 * 33]), ("tiles", 0, [3 * 1024, 1024, 0, 5 * 1024]), ("tiles", 0, [2 * 1024])]
        for variant in (0, 1):
            tracer.set_kernel_variant(variant)
            for kind, width, lengths in cases:
                bufs = [make(n) for n in lengths]
                jobs = [(n, b[0].data_ptr() if n else 0, b[1].data_ptr() if n else 0, b[2].data_ptr() if n else 0) for n, b in zip(lengths, bufs)]
                torch.cuda.synchronize()
                tracer.trace_image_batch_device(jobs, width, tile_world=3 if kind == "tiles" else 1)
                for n, b in zip(lengths, bufs):
                    if n:
                        tracer.trace_device(n, b[0].data_ptr(), b[1].data_ptr(), b[3].data_ptr())
                tracer.synchronize()
                for n, b in zip(lengths, bufs):
                    assert bool((b[2][n:] == -3.0).all()), (variant, kind, width, n)
                    if n:
                        assert torch.equal(b[2][:n].view(torch.int32), b[3].view(torch.int32)), (variant, kind, width, n)
        g0 = make(70 * 3)
        with pytest.raises(RtamdError):
            tracer.trace_image_batch_device([(70 * 3 - 1, g0[0].data_ptr(), g0[1].data_ptr(), g0[2].data_ptr())], 70)      # not whole rows
        with pytest.raises(RtamdError):
            tracer.trace_image_batch_device([(70 * 3, g0[0].data_ptr(), g0[1].data_ptr(), g0[2].data_ptr())], 0)          # no width
        with pytest.raises(RtamdError):
            tracer.trace_image_batch_device([(70 * 3, g0[0].data_ptr(), g0[1].data_ptr(), g0[2].data_ptr())], 0, tile_world=2)   # not whole tiles
    finally:
        tracer.set_kernel_variant(0)
        tracer.set_option("dyn_thresh", 0)
        tracer.set_option("dyn_incoherent", 32)
    with pytest.raises(RtamdError):
        tracer.set_option("dyn_incoherent", 1)
