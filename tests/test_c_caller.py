"""The C-ABI from a plain C99 program (tests/c_caller/drop_in.c): the drop-in boundary as a non-Python, non-C++ caller sees
it — header compiles as strict C, struct layouts are the library's, errors come back as status + text, and the pixels are
the reference's.  CPU: the source compiles with -std=c99 -pedantic -Werror and the binary reports "no GPU" (exit 3, no
fallback).  GPU: the f16 and teapot fixtures through the binary equal the llvmpipe outputs."""
import os
import struct
import subprocess

import numpy as np
import pytest

from common import bits, load_golden, reference_colour, rgb_err_255

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "c_caller", "drop_in.c")
BIN = os.path.join(REPO, "tests", "c_caller", "drop_in")         # built by __graft_entry__.build(), travels to the GPU box
LIB_DIR = os.path.join(REPO, "raytracer_amd", "lib")


def _compile(exe):
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(REPO, "include"), SRC, "-L" + LIB_DIR, "-lrtamd",
                    "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_c99_caller_compiles_strictly_and_fails_loudly_without_a_gpu(tmp_path):
    import torch
    exe = _compile(str(tmp_path / "drop_in"))
    assert os.path.exists(BIN), "tests/c_caller/drop_in not built: run __graft_entry__.build()"
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run is test_c_caller_renders_the_reference_pixels")
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 3 and "no GPU" in p.stdout, (p.returncode, p.stdout, p.stderr)


def _write_scene(path, g):
    h, w = g["dir"].shape[:2]
    sc = g["scene"]
    tex = [g.get("tex0"), g.get("tex1")]
    with open(path, "wb") as f:
        f.write(struct.pack("<12I", 0x314E4353, len(sc["nodes"]), len(sc["triangles"]), len(sc["positions"]), int(sc["tex_coords"] is not None),
                            int(sc["normals"] is not None), w, h, *(t.shape[1] if t is not None else 0 for t in tex), *(t.shape[0] if t is not None else 0 for t in tex)))
        for k in ("nodes", "triangles", "positions", "tex_coords", "normals"):
            if sc[k] is not None:
                f.write(np.ascontiguousarray(sc[k]).tobytes())
        f.write(np.ascontiguousarray(g["dir"], np.float32).tobytes())
        f.write(np.ascontiguousarray(g["org"], np.float32).tobytes())
        for t in tex:
            if t is not None:
                f.write(np.ascontiguousarray(t, np.uint8).tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["f16", "teapot"])
def test_c_caller_renders_the_reference_pixels(tmp_path, name):
    # the binary __graft_entry__.build() made travels with the tree; a tree without it gets a fresh one (a 0.2 s gcc run, no HIP)
    exe = BIN if os.path.exists(BIN) else _compile(str(tmp_path / "drop_in"))
    g = load_golden(name)
    h, w = g["dir"].shape[:2]
    n = w * h
    scene, out = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    _write_scene(scene, g)
    p = subprocess.run([exe, scene, out], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.startswith("ok:"), (p.returncode, p.stdout, p.stderr)
    raw = np.fromfile(out, np.uint8)
    assert raw.size == n * (16 + 4 + 4 + 4 + 16)
    o = 0
    rgba = raw[o:o + 16 * n].view(np.float32).reshape(n, 4); o += 16 * n
    hit_id = raw[o:o + 4 * n].view(np.uint32); o += 4 * n
    hit_t = raw[o:o + 4 * n].view(np.float32); o += 4 * n
    rgba8 = raw[o:o + 4 * n].view(np.uint32); o += 4 * n
    hits = raw[o:o + 16 * n].view(np.float32).reshape(n, 4)
    want = g["b_trace"].reshape(-1, 4)
    assert np.array_equal(bits(hits), bits(want))                                    # ray_trace.frag on llvmpipe, every record
    hit = want[:, 0] < 1000.0
    assert np.array_equal(hit_id != 0xFFFFFFFF, hit) and np.array_equal(hit_id[hit], bits(want[:, 3])[hit]) and np.array_equal(bits(hit_t)[hit], bits(want[:, 0])[hit])
    assert rgb_err_255(rgba, reference_colour(g)).max() <= 1.0                        # ray_trace_and_shade.frag on llvmpipe, every pixel
    # the 8-bit back buffer is the GL unorm conversion of the float pixel (util/display.frag)
    c = np.clip(rgba.astype(np.float64), 0, 1)
    q = np.floor(c * 255 + 0.5).astype(np.uint32)
    got = np.stack([(rgba8 >> s) & 255 for s in (0, 8, 16, 24)], 1)
    assert np.abs(got.astype(np.int64) - q.astype(np.int64)).max() <= 1 and (got == q).mean() > 0.999


@pytest.mark.gpu
@pytest.mark.parametrize("args,transport", [(["--gpus", "1"], "RCCL"), (["--gpus", "3", "--copy"], "copy"), (["--gpus", "1", "--copy"], "copy")])
def test_c_caller_multi_gpu_entry_points_reproduce_the_single_gpu_frame(tmp_path, args, transport):
    """The one-process multi-GPU entry points (rtamd_multi_*) from the C99 program: `--gpus 1` goes through RCCL itself
    (ncclCommInitAll + ncclGather with a world of one: all that a one-GPU box can show of the real collective);
    `--gpus 3 --copy` runs the N > 1 logic — interleaved tiles per context, two alternating sets of share buffers, assembly
    on the root — with three contexts on GPU 0 and the copy transport.  The program itself compares every pixel of both
    gather formats with the single context's frame and exits non-zero on a difference."""
    exe = BIN if os.path.exists(BIN) else _compile(str(tmp_path / "drop_in"))
    g = load_golden("f16")
    scene, out = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    _write_scene(scene, g)
    p = subprocess.run([exe, scene, out] + args, capture_output=True, text=True, timeout=200)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert f"multi ok: {args[1]} GPU(s), {transport} transport" in p.stdout, p.stdout
