"""rtamd_multi_* — one process, several GPUs, the collective inside the library — through the Python binding: batches
enqueued back to back (two alternating sets of share buffers, gather and assembly of one batch beside the rendering of the
next) give, frame for frame, the bits one GPU renders alone; with RCCL (a world of one: what a one-GPU box can show) and
with the copy transport and 2 / 3 / 8 contexts on GPU 0 (the N > 1 tile split, ragged edge tiles, ranks that own one tile
fewer than others)."""
import math

import numpy as np
import pytest

from common import load_golden
from raytracer_amd import host, scenes
from raytracer_amd.renderer import GATHER_RGBA8, GATHER_RGBA32F, TRANSPORT_COPY, TRANSPORT_RCCL, HipRayTracer, MultiGpuRayTracer, RtamdError

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices,transport", [([0], TRANSPORT_RCCL), ([0, 0], TRANSPORT_COPY), ([0, 0, 0], TRANSPORT_COPY), ([0] * 8, TRANSPORT_COPY)])
def test_multi_gpu_batches_equal_single_gpu_frames(devices, transport):
    import torch
    g = load_golden("f16")
    w, h = 200, 117                      # 7 x 4 = 28 tiles: ragged right and top edges, 28 % 3 and 28 % 8 != 0
    eye, d, fov = scenes.framing_camera(g["positions"], 0.55)
    cams = [host.Camera(w / h, fov=fov * (1.0 + 0.07 * k), position=eye * np.float32(1.0 + 0.05 * k), direction=d) for k in range(7)]
    frames = [{"camera": (c.inv_proj_view(), c.position)} for c in cams]
    single = HipRayTracer(0)
    multi = MultiGpuRayTracer(devices, transport)
    try:
        for t in (single, multi):
            t.upload_scene(g["scene"])
            t.upload_texture(0, g["tex0"]); t.upload_texture(1, g["tex1"])
        dev = torch.device("cuda", 0)
        want32 = torch.zeros((len(frames), w * h, 4), dtype=torch.float32, device=dev)
        want8 = torch.zeros((len(frames), w * h), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for k, fr in enumerate(frames):
            single.render_device(w, h, camera=fr["camera"], has_tex_coords=1, has_normals=1, rgba_ptr=want32[k].data_ptr(), rgba8_ptr=want8[k].data_ptr())
        single.synchronize()
        assert len({int(x) for x in want8.to(torch.int64).sum(1).tolist()}) == len(frames)      # the frames differ
        for fmt, want in ((GATHER_RGBA8, want8), (GATHER_RGBA32F, want32)):
            # five calls in a row without waiting: 3 + 2 + 1 + 1 frames, then all 7 at once
            got = torch.full_like(want, 77)
            again = torch.full_like(want, 77)
            torch.cuda.synchronize()
            step = want[0].numel() * want.element_size()
            k = 0
            for nf in (3, 2, 1, 1):
                multi.render_batch_device(w, h, frames[k:k + nf], got.data_ptr() + k * step, fmt, has_tex_coords=1, has_normals=1)
                k += nf
            multi.render_batch_device(w, h, frames, again.data_ptr(), fmt, has_tex_coords=1, has_normals=1)
            multi.synchronize()
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), fmt
            assert torch.equal(again.view(torch.int32), want.view(torch.int32)), fmt
            gms, ams = multi.last_gather_ms()
            assert gms >= 0.0 and ams >= 0.0
        # host rays (the fixture's own 160x120 arrays) through the synchronous form
        hh, ww = g["dir"].shape[:2]
        r = single.render(ww, hh, dir=g["dir"], org=g["org"], has_tex_coords=1, has_normals=1, want=("rgba", "rgba8"))
        out = multi.render_batch(ww, hh, [{"dir": g["dir"], "org": g["org"]}] * 2, GATHER_RGBA8, has_tex_coords=1, has_normals=1)
        assert np.array_equal(out[0], r["rgba8"]) and np.array_equal(out[1], r["rgba8"])
        out = multi.render_batch(ww, hh, [{"dir": g["dir"], "org": g["org"]}], GATHER_RGBA32F, has_tex_coords=1, has_normals=1)
        assert np.array_equal(out[0].view(np.uint32), r["rgba"].view(np.uint32))
        with pytest.raises(RtamdError):
            multi.render_batch(ww, hh, [], GATHER_RGBA8)
    finally:
        multi.close()
        single.close()


def test_multi_gpu_create_rejects_bad_device_lists():
    with pytest.raises(RtamdError) as e:
        MultiGpuRayTracer([0, 0], TRANSPORT_RCCL)            # RCCL takes one rank per GPU
    assert "listed twice" in str(e.value)
    with pytest.raises(RtamdError):
        MultiGpuRayTracer([], TRANSPORT_COPY)
    with pytest.raises(RtamdError):
        MultiGpuRayTracer([99], TRANSPORT_COPY)
