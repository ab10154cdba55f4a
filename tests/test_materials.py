"""Per-material shading (include/rtamd.h rtamd_upload_materials).

The reference parses Ka / Kd / Ks / Ns / map_Kd (material.rs:5-16) but its shader hard-codes
AMBIENT / DIFFUSE / SPECULAR / SPEC_POW and two textures (ray_trace_and_shade.frag:184-187, 21),
so a reference output exists only for the table that holds those constants: there the extension
must reproduce the default path bit for bit (and with it the llvmpipe fixtures).  Any other table
is checked against oracle A's restatement of the same expressions: parity unpinned by the reference."""
import numpy as np
import pytest

from common import bits, load_golden, reference_colour, rgb_err_255
from raytracer_amd import host, scenes
from raytracer_amd.renderer import SHADING_MATERIAL_DTYPE, reference_material


def _table(entries):
    return np.array([tuple(e) for e in entries], dtype=SHADING_MATERIAL_DTYPE)


def _four_material_scene():
    """Four walls of a box seen from inside, one material each, with uvs and normals."""
    tri, pos, uv, nrm, cam = scenes.atrium()
    tri = np.array(tri, copy=True)
    c = pos[tri[:, :3].astype(np.int64)].mean(axis=1)
    tri[:, 3] = (c[:, 0] > np.median(c[:, 0])).astype(np.uint32) * 2 + (c[:, 2] > np.median(c[:, 2])).astype(np.uint32)
    m = host.Model.from_arrays(tri, pos, uv, nrm)
    m.build_bvh()
    return m.scene_arrays(), cam


def _textures(rng, n):
    return {k: rng.integers(0, 256, (8 + 3 * k, 5 + 2 * k, 3), dtype=np.uint8) for k in range(n)}


def test_oracle_reference_constants_table_equals_the_shader(oracle_a):
    g = load_golden("f16")
    a = oracle_a.render(g["scene"], g["dir"], g["org"], int(g["has_tex_coords"]), int(g["has_normals"]), g.get("tex0"), g.get("tex1"))
    m = oracle_a.render_materials(g["scene"], g["dir"], g["org"], int(g["has_tex_coords"]), int(g["has_normals"]),
                                  _table([reference_material(0), reference_material(1)]), {0: g["tex0"], 1: g["tex1"]})
    for k in ("rgba", "hit_id", "hit_t", "counts", "shadow"):
        assert np.array_equal(a[k].view(np.uint8), m[k].view(np.uint8)), k
    with pytest.raises(ValueError):
        oracle_a.render_materials(g["scene"], g["dir"], g["org"], 1, 1, _table([reference_material(0)]), {})


def test_oracle_materials_change_what_they_should(oracle_a):
    g = load_golden("f16")
    args = (g["scene"], g["dir"], g["org"], int(g["has_tex_coords"]), int(g["has_normals"]))
    base = oracle_a.render_materials(*args, _table([reference_material(0), reference_material(1)]), {0: g["tex0"], 1: g["tex1"]})
    black = oracle_a.render_materials(*args, _table([((0, 0, 0), (0, 0, 0), (0, 0, 0), 30.0, 0)] * 2), {0: g["tex0"], 1: g["tex1"]})
    hit = base["hit_id"] != 0xFFFFFFFF
    assert np.all(black["rgba"][hit, :3] == 0) and np.array_equal(black["rgba"][~hit], base["rgba"][~hit])
    assert np.array_equal(black["hit_id"], base["hit_id"])
    white = oracle_a.render_materials(*args, _table([reference_material(-1), reference_material(5)]), {0: g["tex0"]})
    assert not np.array_equal(white["rgba"][hit], base["rgba"][hit])     # untextured: albedo 1


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
def test_reference_constants_table_is_bit_identical_to_the_default_path(tracer, variant):
    g = load_golden("f16")
    h, w = g["dir"].shape[:2]
    tracer.upload_scene(g["scene"])
    tracer.upload_texture(0, g["tex0"]); tracer.upload_texture(1, g["tex1"])
    kw = dict(dir=g["dir"], org=g["org"], has_tex_coords=int(g["has_tex_coords"]), has_normals=int(g["has_normals"]), want=("rgba", "hit_id", "hit_t", "counts"))
    try:
        tracer.set_kernel_variant(variant)
        base = tracer.render(w, h, **kw)
        tracer.upload_materials([reference_material(0), reference_material(1)])
        mat = tracer.render(w, h, **kw)
        for k in ("rgba", "hit_id", "hit_t", "counts"):
            assert np.array_equal(base[k].view(np.uint8), mat[k].view(np.uint8)), k
        assert rgb_err_255(mat["rgba"], reference_colour(g)).max() <= 1.0      # every pixel, against the reference shader
        tracer.upload_materials(None)
        again = tracer.render(w, h, **kw)
        assert np.array_equal(base["rgba"].view(np.uint8), again["rgba"].view(np.uint8))
    finally:
        tracer.set_kernel_variant(0)
        tracer.upload_materials(None)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
def test_four_materials_four_textures_match_oracle(tracer, oracle_a, variant):
    import math
    sc, cam = _four_material_scene()
    rng = np.random.default_rng(5)
    tex = _textures(rng, 16)
    table = _table([((0.3, 0.2, 0.1), (1.0, 0.9, 0.8), (1.0, 1.0, 1.0), 10.0, 7),
                    ((0.5, 0.5, 0.5), (0.8, 0.8, 0.8), (0.4, 0.4, 0.4), 30.0, 15),
                    ((0.05, 0.6, 0.9), (0.0, 0.25, 1.5), (0.2, 0.0, 0.7), 200.0, -1),
                    ((1.0, 1.0, 1.0), (0.5, 0.5, 0.5), (0.0, 0.0, 0.0), 1.0, 2)])
    d = np.asarray(cam["direction"], np.float32); d /= np.linalg.norm(d)
    camera = host.Camera(16 / 9, fov=math.radians(cam["fov_deg"]), position=cam["position"], direction=d)
    W, H = 160, 90
    dirs, orgs = camera.primary_rays(W, H)
    tracer.upload_scene(sc)
    for k, t in tex.items():
        if k != 9:                                   # slot 9 stays empty (no material uses it)
            tracer.upload_texture(k, t)
    try:
        tracer.set_kernel_variant(variant)
        tracer.upload_materials(table)
        r = tracer.render(W, H, dir=dirs, org=orgs, has_tex_coords=1, has_normals=1)
        a = oracle_a.render_materials(sc, dirs, orgs, 1, 1, table, tex, nthreads=8)
        assert (a["hit_id"] != 0xFFFFFFFF).mean() > 0.5 and len(np.unique(sc["triangles"]["mat_idx"][a["hit_id"][a["hit_id"] != 0xFFFFFFFF]])) == 4
        assert np.array_equal(a["hit_id"], r["hit_id"]) and np.array_equal(bits(a["hit_t"]), bits(r["hit_t"]))
        assert rgb_err_255(a["rgba"], r["rgba"]).max() <= 0.05
        # a texture uploaded after the table is seen by the next frame
        tex[7] = rng.integers(0, 256, (4, 4, 3), dtype=np.uint8)
        tracer.upload_texture(7, tex[7])
        r2 = tracer.render(W, H, dir=dirs, org=orgs, has_tex_coords=1, has_normals=1)
        a2 = oracle_a.render_materials(sc, dirs, orgs, 1, 1, table, tex, nthreads=8)
        assert rgb_err_255(a2["rgba"], r2["rgba"]).max() <= 0.05 and not np.array_equal(r["rgba"], r2["rgba"])
    finally:
        tracer.set_kernel_variant(0)
        tracer.upload_materials(None)


@pytest.mark.gpu
def test_material_table_errors(tracer):
    from raytracer_amd.renderer import RtamdError
    sc, _ = _four_material_scene()
    tracer.upload_scene(sc)
    with pytest.raises(RtamdError) as e:
        tracer.upload_materials([reference_material(0)] * 3)        # the scene uses material index 3
    assert e.value.code == -1 and "material index 3" in str(e.value)
    with pytest.raises(RtamdError):
        tracer.upload_materials([reference_material(16)] * 4)
    with pytest.raises(RtamdError):
        tracer.upload_texture(16, np.zeros((2, 2, 3), np.uint8))
    tracer.upload_materials([reference_material(-1)] * 4)
    g = load_golden("cube")
    tracer.upload_scene(g["scene"])                                 # a new scene drops the table
    white = np.full((1, 1, 3), 255, np.uint8)                       # the tracer fixture is shared: earlier tests left their textures
    tracer.upload_texture(0, white); tracer.upload_texture(1, white)
    h, w = g["dir"].shape[:2]
    r = tracer.render(w, h, dir=g["dir"], org=g["org"], has_tex_coords=0, has_normals=int(g["has_normals"]))
    assert rgb_err_255(r["rgba"], g["b_fused"].reshape(-1, 4)).max() <= 1.0
