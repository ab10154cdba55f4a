"""The two shared libraries load and export every symbol their headers declare;
without a GPU the device library fails loudly (no CPU fallback).  CPU only."""
import ctypes
import os
import re

import pytest

from raytracer_amd import _ffi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtamd_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header,path", [("rtamd.h", _ffi.DEVICE_LIB_PATH), ("rtamd_host.h", _ffi.HOST_LIB_PATH)])
def test_library_exports_every_declared_symbol(header, path):
    assert os.path.exists(path), f"{path} not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    names = declared_functions(header)
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/{header} but not exported: {missing}"


def test_device_library_is_gfx950_code_object():
    blob = open(_ffi.DEVICE_LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_persistent" in blob


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _ffi.device_lib()
    h = ctypes.c_void_p()
    rc = lib.rtamd_context_create(0, ctypes.byref(h))
    assert rc == -5 and not h.value            # RTAMD_ERR_DEVICE
    assert b"HIP device" in lib.rtamd_last_error(None)
    from raytracer_amd.renderer import HipRayTracer, RtamdError
    with pytest.raises(RtamdError):
        HipRayTracer(0)


def test_null_context_is_rejected():
    lib = _ffi.device_lib()
    assert lib.rtamd_synchronize(None) == -1
    assert lib.rtamd_upload_scene(None, None) == -1
    assert lib.rtamd_tiles_per_rank(1920, 1080, 8) == 255    # 60 x 34 tiles of 32 x 32
    assert lib.rtamd_tiles_per_rank(1920, 1080, 0) == 0
    assert lib.rtamd_version().startswith(b"rtamd")


def test_product_never_references_the_oracle():
    """The product path must not import, link or load anything under oracle/."""
    bad = []
    for root in ("raytracer_amd",):
        for dp, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                    text = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"oracle_a|liboracle|glsl_ref|from oracle|import oracle|oracles", text):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_abi_version_of_header_library_and_bindings_agree():
    import re
    from raytracer_amd import _ffi
    hdr = open(os.path.join(REPO, "include", "rtamd.h")).read()
    v = int(re.search(r"#define RTAMD_ABI_VERSION (\d+)", hdr).group(1))
    lib = ctypes.CDLL(_ffi.DEVICE_LIB_PATH)
    lib.rtamd_abi_version.restype = ctypes.c_uint32
    assert v == _ffi.ABI_VERSION == lib.rtamd_abi_version()


def test_ctypes_struct_layouts_equal_the_headers(tmp_path):
    """Every struct the Python bindings declare (raytracer_amd/_ffi.py, renderer.SHADING_MATERIAL_DTYPE) has the size and the
    field offsets a C compiler gives the declaration in include/rtamd.h."""
    import subprocess
    from raytracer_amd import _ffi, renderer
    pairs = {"rtamd_scene_desc": _ffi.SceneDesc, "rtamd_frame_desc": _ffi.FrameDesc, "rtamd_frame_out": _ffi.FrameOut,
             "rtamd_dispatch_desc": _ffi.DispatchDesc, "rtamd_dispatch_out": _ffi.DispatchOut, "rtamd_scene_info": _ffi.SceneInfo,
             "rtamd_trace_job": _ffi.TraceJob, "rtamd_multibounce_desc": _ffi.MultiBounceDesc, "rtamd_multibounce_lane": _ffi.MultiBounceLane}
    lines = []
    for cname, cls in pairs.items():
        lines.append(f'printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('printf("\\n");')
    mat_fields = renderer.SHADING_MATERIAL_DTYPE.names
    lines.append('printf("rtamd_shading_material %zu", sizeof(rtamd_shading_material));')
    lines += [f'printf(" %zu", offsetof(rtamd_shading_material, {f}));' for f in mat_fields] + ['printf("\\n");']
    src = tmp_path / "layout.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "rtamd.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0;\n}\n")
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(REPO, "include"), str(src), "-o", exe], check=True)
    out = {ln.split()[0]: [int(x) for x in ln.split()[1:]] for ln in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines()}
    for cname, cls in pairs.items():
        want = [ctypes.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert out[cname] == want, (cname, out[cname], want)
    dt = renderer.SHADING_MATERIAL_DTYPE
    assert out["rtamd_shading_material"] == [dt.itemsize] + [dt.fields[f][1] for f in mat_fields]
