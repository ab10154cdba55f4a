"""GPU parity (through the C-ABI, librtamd.so): HIP kernels vs the committed
llvmpipe fixtures (bit-exact hit records, RGB <= 1/255) and vs oracle A on the
same seeded inputs."""
import numpy as np
import pytest

from common import DISPATCH_KEYS, GOLDEN_SCENES, bits, check_dispatch, load_golden, reference_colour, rgb_err_255

pytestmark = pytest.mark.gpu


def _upload(tracer, g):
    tracer.upload_scene(g["scene"])
    white = np.full((1, 1, 3), 255, np.uint8)
    tracer.upload_texture(0, g["tex0"] if "tex0" in g else white)
    tracer.upload_texture(1, g["tex1"] if "tex1" in g else white)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("name", GOLDEN_SCENES)
def test_trace_matches_reference_shader_bit_exact(tracer, name, variant):
    g = load_golden(name)
    _upload(tracer, g)
    tracer.set_kernel_variant(variant)
    try:
        hits = tracer.trace(g["org"], g["dir"])
        assert np.array_equal(bits(hits), bits(g["b_trace"].reshape(-1, 4)))
        hits = tracer.trace(g["rand_org"], g["rand_dir"])
        assert np.array_equal(bits(hits), bits(g["b_trace_rand"].reshape(-1, 4)))
    finally:
        tracer.set_kernel_variant(0)


def test_trace_special_values(tracer):
    g = load_golden("special")
    tracer.upload_scene({"nodes": g["nodes"], "triangles": g["triangles"], "positions": g["positions"]})
    hits = tracer.trace(g["org"], g["dir"])
    assert np.array_equal(bits(hits), bits(g["b_trace"].reshape(-1, 4)))


@pytest.mark.parametrize("name", ["f16", "testobj", "atrium"])
def test_textured_colour_with_the_reference_executors_filter(tracer, name):
    """Option "texture_filter_8bit": the kernels filter as llvmpipe does (restated bit for bit on its probed samples,
    tests/test_oracle_golden.py), and the textured fixtures then agree with llvmpipe to 0.02/255 on every pixel instead of
    0.9/255 -- the library's texture-coordinate interpolation, v flip, wrapping and slot choice are the reference's."""
    from common import atrium_golden
    if name == "atrium":
        g = atrium_golden()
        tracer.upload_scene(g["scene"])
        tracer.upload_texture(0, g["tex0"]); tracer.upload_texture(1, g["tex1"])
        w, h, flags, want = g["w"], g["h"], (1, 1), g["b_fused_uniform"]
    else:
        g = load_golden(name)
        _upload(tracer, g)
        h, w = g["dir"].shape[:2]
        flags, want = (int(g["has_tex_coords"]), int(g["has_normals"])), reference_colour(g)
    tracer.set_option("texture_filter_8bit", 1)
    try:
        r = tracer.render(w, h, dir=g["dir"], org=g["org"], has_tex_coords=flags[0], has_normals=flags[1], want=("rgba",))
    finally:
        tracer.set_option("texture_filter_8bit", 0)
    err = rgb_err_255(r["rgba"].reshape(-1, 4), want.reshape(-1, 4))
    assert err.max() <= 0.02, float(err.max())


@pytest.mark.parametrize("variant", [0, 1])
def test_rays_on_triangle_edges(tracer, variant):
    """golden_edge: the decisions of intersectTriangle within an ulp of an edge or a vertex, against llvmpipe's records and
    colours; half of the rays separate `u + v` as written from the compiler's f*(du + dv) (tests/test_oracle_golden.py)."""
    g = load_golden("edge")
    tracer.upload_scene({"nodes": g["nodes"], "triangles": g["triangles"], "positions": g["positions"]})
    tracer.set_kernel_variant(variant)
    try:
        hits = tracer.trace(g["org"], g["dir"])
        h, w = g["dir"].shape[:2]
        r = tracer.render(w, h, dir=g["dir"], org=g["org"], has_tex_coords=0, has_normals=0, want=("rgba", "hit_id", "hit_t"))
    finally:
        tracer.set_kernel_variant(0)
    want = g["b_trace"].reshape(-1, 4)
    assert np.array_equal(bits(hits), bits(want))
    hit = want[:, 0] < 1000
    assert np.array_equal(bits(r["hit_t"].reshape(-1)[hit]), bits(want[hit, 0]))
    assert np.array_equal(r["hit_id"].reshape(-1)[hit], want[hit, 3].view(np.uint32))
    assert rgb_err_255(r["rgba"].reshape(-1, 4), g["b_fused"].reshape(-1, 4)).max() <= 1.0


@pytest.mark.parametrize("anyhit", [0, 1])
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("name", GOLDEN_SCENES)
def test_fused_render_matches_reference_shader(tracer, oracle_a, name, variant, anyhit):
    """anyhit=0: shadow rays traverse to the end like the reference, so the per-pixel test
    counters equal the oracle's too; anyhit=1 (the default): shadow rays stop at the first
    occluder, every output still identical, counters can only be smaller."""
    g = load_golden(name)
    h, w = g["dir"].shape[:2]
    _upload(tracer, g)
    tracer.set_kernel_variant(variant)
    tracer.set_option("shadow_anyhit", anyhit)
    try:
        r = tracer.render(w, h, dir=g["dir"], org=g["org"], has_tex_coords=int(g["has_tex_coords"]),
                          has_normals=int(g["has_normals"]), want=("rgba", "hit_id", "hit_t", "counts"))
    finally:
        tracer.set_kernel_variant(0)
        tracer.set_option("shadow_anyhit", 1)
    a = oracle_a.render(g["scene"], g["dir"], g["org"], int(g["has_tex_coords"]), int(g["has_normals"]), g.get("tex0"), g.get("tex1"))
    # hit buffers: identical to oracle A on every pixel, identical to the reference trace shader on hit pixels
    assert np.array_equal(r["hit_id"], a["hit_id"])
    assert np.array_equal(bits(r["hit_t"]), bits(a["hit_t"]))
    if anyhit:
        assert np.all(r["counts"] <= a["counts"]) and np.array_equal(r["counts"][~a["shadow"].astype(bool)], a["counts"][~a["shadow"].astype(bool)])
    else:
        assert np.array_equal(r["counts"], a["counts"])  # same boxes and triangles tested => same traversal order
    want_trace = g["b_trace"].reshape(-1, 4)
    hit = r["hit_id"] != 0xFFFFFFFF
    assert np.array_equal(hit, want_trace[:, 0] < 1000.0)
    assert np.array_equal(r["hit_id"][hit], bits(want_trace[:, 3])[hit])
    assert np.array_equal(bits(r["hit_t"])[hit], bits(want_trace[:, 0])[hit])
    # colour against the reference shader on llvmpipe on EVERY pixel: 0.06/255 without textures (the GPU's pow), 1/255 with
    # them (llvmpipe filters in 8 bits; 0.02/255 with option "texture_filter_8bit", tested below), and against oracle A
    assert rgb_err_255(r["rgba"], reference_colour(g)).max() <= (1.0 if int(g["has_tex_coords"]) else 0.06)
    assert rgb_err_255(r["rgba"], a["rgba"]).max() <= 0.05
    assert np.all(r["rgba"][:, 3] == 1.0)


@pytest.mark.parametrize("variant", [0, 1])
def test_headline_scene_matches_reference_shaders(tracer, variant):
    """bench.py's scene against the llvmpipe fixture (tests/golden/golden_atrium.npz): trace records and the fused
    shader's hit buffers bit-exact, colour <= 1/255 on every pixel."""
    from common import atrium_golden
    g = atrium_golden()
    w, h = g["w"], g["h"]
    tracer.upload_scene(g["scene"])
    tracer.upload_texture(0, g["tex0"]); tracer.upload_texture(1, g["tex1"])
    try:
        tracer.set_kernel_variant(variant)
        t = tracer.trace(g["org"].reshape(-1, 3), g["dir"].reshape(-1, 3))
        r = tracer.render(w, h, dir=g["dir"], org=g["org"], has_tex_coords=1, has_normals=1)
    finally:
        tracer.set_kernel_variant(0)
    assert np.array_equal(bits(t), bits(g["b_trace"]))
    hit = r["hit_id"] != 0xFFFFFFFF
    assert np.array_equal(hit, g["b_trace"][:, 0] < 1000.0)
    assert np.array_equal(r["hit_id"][hit], bits(g["b_trace"][:, 3])[hit]) and np.array_equal(bits(r["hit_t"])[hit], bits(g["b_trace"][:, 0])[hit])
    assert rgb_err_255(r["rgba"], g["b_fused_uniform"]).max() <= 1.0      # every pixel


@pytest.mark.parametrize("variant", [0, 1])
def test_triangle_soup_matches_reference_shaders(tracer, variant):
    """tests/golden/golden_soup.npz: 100 000-triangle soup, camera frame + 8192 random rays against llvmpipe."""
    from common import soup_golden
    g = soup_golden()
    tracer.upload_scene(g["scene"])
    white = np.full((1, 1, 3), 255, np.uint8)
    tracer.upload_texture(0, white); tracer.upload_texture(1, white)
    try:
        tracer.set_kernel_variant(variant)
        assert np.array_equal(bits(tracer.trace(g["org"].reshape(-1, 3), g["dir"].reshape(-1, 3))), bits(g["b_trace"]))
        assert np.array_equal(bits(tracer.trace(g["rand_org"], g["rand_dir"])), bits(g["b_trace_rand"]))
        r = tracer.render(g["w"], g["h"], dir=g["dir"], org=g["org"], has_tex_coords=0, has_normals=0)
    finally:
        tracer.set_kernel_variant(0)
    hit = r["hit_id"] != 0xFFFFFFFF
    assert np.array_equal(hit, g["b_trace"][:, 0] < 1000.0) and np.array_equal(r["hit_id"][hit], bits(g["b_trace"][:, 3])[hit])
    assert rgb_err_255(r["rgba"], g["b_fused"]).max() < 0.01


def test_ray_creation_matches_reference_shader(tracer):
    """rtamd_generate_rays against the llvmpipe runs of ray_create.frag (three cameras)."""
    from common import check_raycreate
    from raytracer_amd import host
    def gen(w, h, inv, near, far):
        assert (np.float32(near), np.float32(far)) == (np.float32(host.NEAR), np.float32(host.FAR))
        return tracer.generate_rays(w, h, inv)
    check_raycreate(gen)


@pytest.mark.parametrize("fixture", ["dispatch", "dispatch2"])
def test_ray_dispatcher_matches_reference_shader(tracer, oracle_a, fixture):
    """k_dispatch_rays vs ray_dispatcher.frag on llvmpipe (two fixtures: bunny frame / seeded noise, f16 frame / the
    reference's blue noise) and vs oracle A, through the host entry point; then ragged sizes (partial last block, sizes
    below one block) vs oracle A bit for bit."""
    g = load_golden(fixture)
    tracer.upload_noise_texture(g["noise"])
    args = (g["light_pos"], g["camera_pos"], g["noise_offset_scale"])
    r = tracer.dispatch_rays(g["position"], g["normal_mat"], *args)
    check_dispatch(r, g)
    a = oracle_a.dispatch(g["position"], g["normal_mat"], g["noise"], *args)
    for k in DISPATCH_KEYS:
        assert np.array_equal(bits(r[k]), bits(a[k])), k          # same float bilinear on both sides
    for h, w in ((7, 150), (1, 1), (3, 85), (2, 256)):
        p, nm = g["position"][:h, :w], g["normal_mat"][:h, :w]
        r = tracer.dispatch_rays(p, nm, *args)
        a = oracle_a.dispatch(p, nm, g["noise"], *args)
        for k in DISPATCH_KEYS:
            assert np.array_equal(bits(r[k]), bits(a[k])), (h, w, k)


def test_ray_dispatcher_device_entry_and_trace_pipeline(tracer, oracle_a):
    """The v0 pipeline of src/main.rs:156-203 on device buffers: dispatcher -> three ray_trace passes
    (shadow, reflection, ambient rays); every hit record equals oracle A's for the same rays, and
    pixels without a surface (NO_RAY) keep the cleared record."""
    import torch
    g = load_golden("dispatch")
    b = load_golden("bunny")
    tracer.upload_scene(b["scene"])
    tracer.upload_noise_texture(g["noise"])
    h, w = g["normal_mat"].shape[:2]
    n = h * w
    dev = torch.device("cuda", 0)
    d_pos = torch.from_numpy(g["position"].reshape(n, 3)).to(dev)
    d_nm = torch.from_numpy(g["normal_mat"].reshape(n, 4)).to(dev)
    outs = {k: torch.full((n, 3), 7.0, dtype=torch.float32, device=dev) for k in DISPATCH_KEYS}
    tracer.dispatch_rays_device(w, h, d_pos.data_ptr(), d_nm.data_ptr(), g["light_pos"], g["camera_pos"], g["noise_offset_scale"],
                                **{k + "_ptr": v.data_ptr() for k, v in outs.items()})
    hits = {}
    for k in DISPATCH_KEYS[1:]:
        hits[k] = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        tracer.trace_device(n, outs["org"].data_ptr(), outs[k].data_ptr(), hits[k].data_ptr())
    tracer.synchronize()
    a = oracle_a.dispatch(g["position"], g["normal_mat"], g["noise"], g["light_pos"], g["camera_pos"], g["noise_offset_scale"])
    surface = (g["normal_mat"][..., 3] != np.float32(1e30)).reshape(-1)
    for k in DISPATCH_KEYS:
        assert np.array_equal(bits(outs[k].cpu().numpy()), bits(a[k].reshape(n, 3))), k
    for k in DISPATCH_KEYS[1:]:
        want = oracle_a.trace(b["scene"], a["org"], a[k])["hits"]
        got = hits[k].cpu().numpy()
        assert np.array_equal(bits(got), bits(want)), k
        assert np.all(got[~surface, 0] == np.float32(1e30)) and np.all(got[~surface, 1:] == 0)
    assert (hits["shadow_dir"].cpu().numpy()[surface, 0] < 1e30).sum() > 100   # some surface points are shadowed


@pytest.mark.parametrize("name", ["bunny", "teapot", "f16"])
def test_surface_from_hits_and_two_bounces_match_oracle(tracer, oracle_a, name):
    """Hit records -> G-buffer (interpolated normals for bunny / f16, flat for teapot) and the repeated
    dispatcher -> trace loop the reference's one-bounce pipeline stops short of: every buffer of two bounces
    (surface, four ray targets, three hit-record sets per bounce) equals oracle A bit for bit."""
    g = load_golden(name)
    d = load_golden("dispatch")
    tracer.upload_scene(g["scene"])
    tracer.upload_noise_texture(d["noise"])
    has_n = int(g["has_normals"])
    h, w = g["dir"].shape[:2]
    org, dirs = g["org"].reshape(-1, 3), g["dir"].reshape(-1, 3)
    hits = tracer.trace(org, dirs)
    assert np.array_equal(bits(hits), bits(g["b_trace"].reshape(-1, 4)))
    light, nos = d["light_pos"], d["noise_offset_scale"]
    cam = org[0]
    for bounce in range(2):
        pos, nm = tracer.surface_from_hits(org, dirs, hits, has_n)
        apos, anm = oracle_a.surface_from_hits(g["scene"], org, dirs, hits, has_n)
        assert np.array_equal(bits(pos), bits(apos)) and np.array_equal(bits(nm), bits(anm)), bounce
        surface = nm[:, 3] != np.float32(1e30)
        assert np.array_equal(surface, hits[:, 0] < np.float32(1e30)) and surface.sum() > (1000 if bounce == 0 else 10)
        r = tracer.dispatch_rays(pos.reshape(h, w, 3), nm.reshape(h, w, 4), light, cam, nos)
        a = oracle_a.dispatch(pos.reshape(h, w, 3), nm.reshape(h, w, 4), d["noise"], light, cam, nos)
        for k in DISPATCH_KEYS:
            assert np.array_equal(bits(r[k]), bits(a[k])), (bounce, k)
        org = r["org"].reshape(-1, 3)
        for k in DISPATCH_KEYS[1:]:
            got = tracer.trace(org, r[k].reshape(-1, 3))
            want = oracle_a.trace(g["scene"], org, r[k].reshape(-1, 3))["hits"]
            assert np.array_equal(bits(got), bits(want)), (bounce, k)
            if k == "reflect_dir":
                next_hits, next_dirs = got, r[k].reshape(-1, 3)
        hits, dirs = next_hits, next_dirs


def test_offset_ray_generation(tracer, oracle_a):
    """Sample offset (0.5, 0.5) is the pixel-centre generator bit for bit; other offsets move every ray
    continuously between its neighbours' directions."""
    import torch
    g = load_golden("bunny")
    tracer.upload_scene(g["scene"])
    cam = __import__("raytracer_amd").host.Camera(4 / 3, position=(0.5, 0.2, 2.0), direction=(0.0, 0.0, -1.0))
    inv = cam.inv_proj_view()
    W, H = 64, 48
    dev = torch.device("cuda", 0)
    out = [torch.zeros((W * H, 3), dtype=torch.float32, device=dev) for _ in range(3)]
    tracer.generate_rays_device(W, H, inv, out[0].data_ptr())
    tracer.generate_rays_offset_device(W, H, inv, 0.5, 0.5, out[1].data_ptr())
    tracer.generate_rays_offset_device(W, H, inv, 1.0, 0.5, out[2].data_ptr())
    tracer.synchronize()
    c, same, right = (o.cpu().numpy().reshape(H, W, 3) for o in out)
    assert np.array_equal(bits(c), bits(same))
    from raytracer_amd.host import FAR, NEAR
    assert np.array_equal(bits(right), bits(oracle_a.raycreate(W, H, inv, NEAR, FAR, offset=(1.0, 0.5))))    # any offset: the oracle's bits
    mid = (c[:, :-1] + c[:, 1:]) / 2
    mid /= np.linalg.norm(mid, axis=-1, keepdims=True)
    assert np.abs(right[:, :-1] - mid).max() < 2e-3
    with pytest.raises(Exception):
        tracer.generate_rays_offset_device(W, H, inv, 1.5, 0.5, out[2].data_ptr())


@pytest.mark.parametrize("w,h", [(37, 19), (1, 1), (8, 65), (131, 3)])
def test_ragged_frame_sizes_and_ray_counts(tracer, oracle_a, w, h):
    """Frames that are not multiples of the 8x8 work item / 32x32 tile, and ray counts that are not
    multiples of 64: every pixel / ray equals oracle A, nothing outside the frame is written."""
    g = load_golden("f16")
    _upload(tracer, g)
    dirs = np.ascontiguousarray(g["dir"][40:40 + h, 60:60 + w]) if h <= 80 and w <= 100 else np.ascontiguousarray(g["dir"].reshape(-1, 3)[:w * h].reshape(h, w, 3))
    orgs = np.broadcast_to(g["org"][0, 0], dirs.shape).copy()
    for variant in (0, 1):
        tracer.set_kernel_variant(variant)
        try:
            r = tracer.render(w, h, dir=dirs, org=orgs, has_tex_coords=1, has_normals=1, want=("rgba", "hit_id", "hit_t"))
            t = tracer.trace(orgs.reshape(-1, 3), dirs.reshape(-1, 3))
        finally:
            tracer.set_kernel_variant(0)
        a = oracle_a.render(g["scene"], dirs, orgs, 1, 1, g["tex0"], g["tex1"])
        assert r["rgba"].shape == (w * h, 4)
        assert np.array_equal(r["hit_id"], a["hit_id"]) and np.array_equal(bits(r["hit_t"]), bits(a["hit_t"]))
        assert rgb_err_255(a["rgba"], r["rgba"]).max() <= 0.05
        assert np.array_equal(bits(t), bits(oracle_a.trace(g["scene"], orgs, dirs)["hits"]))
    # tile shards of a ragged frame: 3 ranks, compact tile buffers re-assemble to the frame
    from raytracer_amd.renderer import assemble_tiles_host
    parts = [tracer.render(w, h, dir=dirs, org=orgs, has_tex_coords=1, has_normals=1, tile_rank=k, tile_world=3)["rgba"] for k in range(3)]
    per = tracer.tiles_per_rank(w, h, 3) * 1024
    frame = assemble_tiles_host([np.concatenate([p, np.zeros((per - len(p), 4), np.float32)]) for p in parts], w, h, 3, 4).reshape(-1, 4)
    assert np.array_equal(bits(frame), bits(r["rgba"]))
