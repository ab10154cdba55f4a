"""rtamd_render_device_static — the entry point that allocates nothing (include/rtamd.h) — against
rtamd_render_device, through the C-ABI, bit for bit: two calls in a row (the second with a smaller
input, so stale contents of the static launch scratch would show), a tile shard with the 8-bit output,
a tree deep enough to use the static stack columns, and the rejection beyond the stated limit.
Collected last (file name): the round's end runs `pytest -x`."""
import numpy as np
import pytest
import torch

from common import load_golden
from raytracer_amd.renderer import TILE_PIXELS, HipRayTracer, RtamdError
from test_gpu_api import _chain_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def own_tracer():
    t = HipRayTracer(0)      # a context of its own: nothing another test left behind (options, scene) matters
    yield t
    t.close()


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _render_both(t, w, h, d_dir, d_org, has_n, has_tc=0, tile_rank=0, tile_world=1, with8=False, counts=False):
    """The same frame through rtamd_render_device and rtamd_render_device_static: dict name -> (existing, static)."""
    n_out = w * h if tile_world == 1 else t.local_tile_count(w, h, tile_rank, tile_world) * TILE_PIXELS
    res = {}
    for which in ("device", "static"):
        out = {"rgba": torch.full((n_out, 4), -7.0, dtype=torch.float32, device="cuda:0"),
               "hit_id": torch.full((n_out,), 12345, dtype=torch.int32, device="cuda:0"),
               "hit_t": torch.full((n_out,), -7.0, dtype=torch.float32, device="cuda:0")}
        if with8:
            out["rgba8"] = torch.full((n_out,), 12345, dtype=torch.int32, device="cuda:0")
        if counts:
            out["counts"] = torch.full((n_out, 2), 12345, dtype=torch.int32, device="cuda:0")
        fn = t.render_device if which == "device" else t.render_device_static
        torch.cuda.synchronize()    # inputs and fills were made on torch's stream; the library launches on the context's own
        fn(w, h, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=has_tc, has_normals=has_n, tile_rank=tile_rank, tile_world=tile_world,
           rgba_ptr=out["rgba"].data_ptr(), hit_id_ptr=out["hit_id"].data_ptr(), hit_t_ptr=out["hit_t"].data_ptr(),
           rgba8_ptr=out["rgba8"].data_ptr() if with8 else 0, counts_ptr=out["counts"].data_ptr() if counts else 0)
        t.synchronize()
        for k, v in out.items():
            res.setdefault(k, []).append(v)
    return res


def _assert_same(res):
    for k, (a, b) in res.items():
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), k


def test_static_entry_point_equals_render_device_on_two_calls_in_a_row(own_tracer):
    t = own_tracer
    g = load_golden("bunny")
    h, w = g["dir"].shape[:2]
    has_n = int(g["has_normals"])
    t.upload_scene(g["scene"])
    assert t.scene_info().max_depth <= 48
    d_dir, d_org = _dev(g["dir"].reshape(-1, 3)), _dev(g["org"].reshape(-1, 3))
    # first call: the whole 160x120 frame (300 work items), with the per-pixel test counters
    r1 = _render_both(t, w, h, d_dir, d_org, has_n, counts=True)
    _assert_same(r1)
    hit = r1["hit_id"][1] != -1
    assert 2000 < int(hit.sum()) < w * h                         # the frame shows the model, and sky
    # the static call allocates nothing: free device memory is the same before and after it
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    t.render_device_static(w, h, d_dir.data_ptr(), d_org.data_ptr(), has_normals=has_n, rgba_ptr=r1["rgba"][1].data_ptr())
    t.synchronize()
    assert torch.cuda.mem_get_info(0)[0] == free0
    # second call, smaller input: the middle 40 rows only (fewer work items than the counters served before)
    rows = slice(40 * w, 80 * w)
    d_dir2, d_org2 = d_dir[rows].contiguous(), d_org[rows].contiguous()
    r2 = _render_both(t, w, 40, d_dir2, d_org2, has_n)
    _assert_same(r2)
    assert torch.equal(r2["hit_id"][1], r1["hit_id"][1][rows]) and torch.equal(r2["rgba"][1], r1["rgba"][1][rows])
    # third: a tile shard (rank 1 of 3) with the display-format output, static call first this time
    r3 = _render_both(t, w, h, d_dir, d_org, has_n, tile_rank=1, tile_world=3, with8=True)
    _assert_same(r3)


def test_static_entry_point_uses_its_stack_columns_and_refuses_deeper_trees(own_tracer):
    t = own_tracer
    depth = 40                                                   # 16 entries in LDS, 24 in the static columns
    t.upload_scene(_chain_scene(depth))
    rng = np.random.default_rng(5)
    n = 64 * 300
    org = np.concatenate([rng.uniform(-3.5, 3.5, (n, 2)), np.full((n, 1), depth + 5.0)], 1).astype(np.float32)
    d = np.tile(np.float32([0, 0, -1]), (n, 1))
    d[1::2] = np.float32([0.01, 0.02, -1]) / np.float32(np.linalg.norm([0.01, 0.02, -1]))
    up = rng.random(n) < 0.25                                    # lanes of one wave at different stack depths
    org[up, 2] = -5.0
    d[up] = np.float32([0, 0, 1])
    r = _render_both(t, n, 1, _dev(d), _dev(org), 0)
    _assert_same(r)
    assert int((r["hit_id"][1] != -1).sum()) > n // 4
    # and again with fewer rays (second call in a row, smaller input)
    m = 64 * 7 + 13
    r = _render_both(t, m, 1, _dev(d[:m]), _dev(org[:m]), 0)
    _assert_same(r)
    t.upload_scene(_chain_scene(49))                             # one level beyond the stated limit
    buf = torch.zeros((64, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    with pytest.raises(RtamdError) as e:
        t.render_device_static(64, 1, _dev(d[:64]).data_ptr(), _dev(org[:64]).data_ptr(), rgba_ptr=buf.data_ptr())
    assert e.value.code == -1 and "48" in str(e.value)
    t.upload_scene(_chain_scene(48))                             # the limit itself is served
    _assert_same(_render_both(t, m, 1, _dev(d[:m]), _dev(org[:m]), 0))


def test_static_calls_of_two_contexts_on_two_streams_do_not_share_stack_entries():
    """All static calls on a GPU share one set of stack columns.  Two contexts with DIFFERENT deep scenes (their trees use
    the columns) issue static calls back to back on different streams without any ordering of the caller's: the library's
    per-device event serialises them on the GPU, so neither pops the other's node references (which would index the wrong
    scene's arrays, out of bounds for the smaller one).  Each result equals the same context's rtamd_render_device."""
    a, b = HipRayTracer(0), HipRayTracer(0)
    try:
        a.upload_scene(_chain_scene(46))
        b.upload_scene(_chain_scene(23))
        sa, sb = torch.cuda.Stream(device="cuda:0"), torch.cuda.Stream(device="cuda:0")
        a.set_stream(sa.cuda_stream); b.set_stream(sb.cuda_stream)
        rng = np.random.default_rng(9)
        n = 64 * 2000
        def rays(depth):
            org = np.concatenate([rng.uniform(-3.5, 3.5, (n, 2)), np.full((n, 1), depth + 5.0)], 1).astype(np.float32)
            d = np.tile(np.float32([0, 0, -1]), (n, 1))
            d[1::2] = np.float32([0.01, 0.02, -1]) / np.float32(np.linalg.norm([0.01, 0.02, -1]))
            up = rng.random(n) < 0.25
            org[up, 2] = -5.0
            d[up] = np.float32([0, 0, 1])
            return _dev(d), _dev(org)
        da, oa = rays(46)
        db, ob = rays(23)
        outs = {k: [torch.full((n, 4), -7.0, dtype=torch.float32, device="cuda:0") for _ in range(4)] for k in "ab"}
        want = {k: torch.full((n, 4), -7.0, dtype=torch.float32, device="cuda:0") for k in "ab"}
        torch.cuda.synchronize()
        a.render_device(n, 1, da.data_ptr(), oa.data_ptr(), rgba_ptr=want["a"].data_ptr())
        b.render_device(n, 1, db.data_ptr(), ob.data_ptr(), rgba_ptr=want["b"].data_ptr())
        for k in range(4):      # interleaved, no synchronisation in between
            a.render_device_static(n, 1, da.data_ptr(), oa.data_ptr(), rgba_ptr=outs["a"][k].data_ptr())
            b.render_device_static(n, 1, db.data_ptr(), ob.data_ptr(), rgba_ptr=outs["b"][k].data_ptr())
        torch.cuda.synchronize()
        for k in range(4):
            assert torch.equal(outs["a"][k].view(torch.int32), want["a"].view(torch.int32)), k
            assert torch.equal(outs["b"][k].view(torch.int32), want["b"].view(torch.int32)), k
        assert float((want["a"][:, 3] == 1.0).float().mean()) == 1.0
    finally:
        a.close(); b.close()
