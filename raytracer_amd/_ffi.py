"""ctypes loaders for the two in-tree shared libraries.

librtamd_host.so  host scene layer (CPU)          include/rtamd_host.h
librtamd.so       C-ABI device library (gfx950)   include/rtamd.h

Both are built by raytracer_amd/csrc/Makefile (see __graft_entry__.build()).
There is deliberately no fallback: a missing library raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
HOST_LIB_PATH = os.path.join(LIB_DIR, "librtamd_host.so")
# RTAMD_DEVICE_LIB selects another build of the same library (kernel A/B experiments, tools/kbench.py)
DEVICE_LIB_PATH = os.environ.get("RTAMD_DEVICE_LIB") or os.path.join(LIB_DIR, "librtamd.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_u8_p = ctypes.POINTER(ctypes.c_uint8)
c_u32_p = ctypes.POINTER(ctypes.c_uint32)


class ModelInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in (
        "triangle_count", "vertex_count", "node_count", "material_count",
        "has_tex_coords", "has_normals", "has_bvh")]


class MaterialRec(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p),
                ("ambient_color", ctypes.c_float * 3), ("diffuse_color", ctypes.c_float * 3),
                ("specular_color", ctypes.c_float * 3), ("transmission_color", ctypes.c_float * 3),
                ("specular_exp", ctypes.c_float), ("transmission", ctypes.c_float), ("optical_density", ctypes.c_float),
                ("ambient_tex", ctypes.c_char_p), ("diffuse_tex", ctypes.c_char_p),
                ("specular_tex", ctypes.c_char_p), ("specular_exp_tex", ctypes.c_char_p)]


class SceneDesc(ctypes.Structure):
    _fields_ = [
        ("nodes", ctypes.c_void_p), ("node_count", ctypes.c_uint32),
        ("triangles", ctypes.c_void_p), ("triangle_count", ctypes.c_uint32),
        ("positions", ctypes.c_void_p), ("vertex_count", ctypes.c_uint32),
        ("tex_coords", ctypes.c_void_p), ("normals", ctypes.c_void_p),
    ]


class FrameDesc(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
        ("dir", ctypes.c_void_p), ("org", ctypes.c_void_p),
        ("inv_proj_view", ctypes.c_float * 16),
        ("near_plane", ctypes.c_float), ("far_plane", ctypes.c_float),
        ("cam_pos", ctypes.c_float * 3),
        ("has_tex_coords", ctypes.c_uint32), ("has_normals", ctypes.c_uint32),
        ("tile_rank", ctypes.c_uint32), ("tile_world", ctypes.c_uint32),
    ]


class FrameOut(ctypes.Structure):
    _fields_ = [
        ("rgba32f", ctypes.c_void_p), ("hit_id", ctypes.c_void_p),
        ("hit_t", ctypes.c_void_p), ("counts", ctypes.c_void_p), ("rgba8", ctypes.c_void_p),
    ]


class DispatchDesc(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
        ("position", ctypes.c_void_p), ("normal_mat", ctypes.c_void_p),
        ("light_pos", ctypes.c_float * 3), ("camera_pos", ctypes.c_float * 3),
        ("noise_offset_scale", ctypes.c_float * 4),
        ("tile_rank", ctypes.c_uint32), ("tile_world", ctypes.c_uint32),
    ]


class MultiBounceDesc(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
        ("inv_proj_view", ctypes.c_float * 16), ("near_plane", ctypes.c_float), ("far_plane", ctypes.c_float),
        ("cam_pos", ctypes.c_float * 3), ("light_pos", ctypes.c_float * 3),
        ("has_normals", ctypes.c_uint32), ("max_depth", ctypes.c_uint32), ("noise_size", ctypes.c_uint32),
        ("tile_rank", ctypes.c_uint32), ("tile_world", ctypes.c_uint32),
    ]


class MultiBounceLane(ctypes.Structure):
    _fields_ = [
        ("cam_org", ctypes.c_void_p), ("dir", ctypes.c_void_p), ("hits", ctypes.c_void_p), ("position", ctypes.c_void_p), ("normal_mat", ctypes.c_void_p),
        ("org", ctypes.c_void_p * 2), ("shadow_dir", ctypes.c_void_p * 2), ("reflect_dir", ctypes.c_void_p * 2), ("ambient_dir", ctypes.c_void_p * 2),
        ("shadow_hits", ctypes.c_void_p),
    ]


class TraceJob(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("org", ctypes.c_void_p), ("dir", ctypes.c_void_p), ("hits", ctypes.c_void_p)]


class DispatchOut(ctypes.Structure):
    _fields_ = [("org", ctypes.c_void_p), ("shadow_dir", ctypes.c_void_p), ("reflect_dir", ctypes.c_void_p), ("ambient_dir", ctypes.c_void_p)]


class SceneInfo(ctypes.Structure):
    _fields_ = [
        ("node_count", ctypes.c_uint32), ("triangle_count", ctypes.c_uint32), ("vertex_count", ctypes.c_uint32),
        ("inner_node_count", ctypes.c_uint32), ("big_leaf_count", ctypes.c_uint32), ("max_depth", ctypes.c_uint32),
        ("device_bytes", ctypes.c_uint64), ("odd_box_count", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
    ]


_host = None
_device = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               f"or `make -C raytracer_amd/csrc host`")
        L = ctypes.CDLL(HOST_LIB_PATH)
        L.rtamd_model_parse_obj.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_size_t]
        L.rtamd_model_parse_obj.restype = ctypes.c_int
        L.rtamd_model_from_arrays.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        L.rtamd_model_from_arrays.restype = ctypes.c_int
        L.rtamd_model_free.argtypes = [ctypes.c_void_p]
        L.rtamd_model_free.restype = None
        L.rtamd_model_build_bvh.argtypes = [ctypes.c_void_p]
        L.rtamd_model_build_bvh.restype = ctypes.c_int
        L.rtamd_model_build_bvh_parallel.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.rtamd_model_build_bvh_parallel.restype = ctypes.c_int
        L.rtamd_material_lib_parse.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_size_t]
        L.rtamd_material_lib_parse.restype = ctypes.c_int
        L.rtamd_material_lib_free.argtypes = [ctypes.c_void_p]
        L.rtamd_material_lib_free.restype = None
        L.rtamd_material_lib_count.argtypes = [ctypes.c_void_p]
        L.rtamd_material_lib_count.restype = ctypes.c_uint32
        L.rtamd_material_lib_get.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(MaterialRec)]
        L.rtamd_material_lib_get.restype = ctypes.c_int
        L.rtamd_model_get_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ModelInfo)]
        L.rtamd_model_get_info.restype = None
        for name in ("rtamd_model_triangles", "rtamd_model_positions", "rtamd_model_tex_coords",
                     "rtamd_model_normals", "rtamd_model_bvh_nodes"):
            getattr(L, name).argtypes = [ctypes.c_void_p]
            getattr(L, name).restype = ctypes.c_void_p
        L.rtamd_model_material_name.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.rtamd_model_material_name.restype = ctypes.c_char_p
        L.rtamd_model_material_lib_name.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.rtamd_model_material_lib_name.restype = ctypes.c_char_p
        L.rtamd_camera_matrices.argtypes = [ctypes.c_float, ctypes.c_float, c_float_p, c_float_p, c_float_p,
                                            c_float_p, c_float_p, c_float_p]
        L.rtamd_camera_matrices.restype = ctypes.c_int
        L.rtamd_camera_direction.argtypes = [ctypes.c_float, ctypes.c_float, c_float_p]
        L.rtamd_camera_direction.restype = None
        L.rtamd_host_generate_primary_rays.argtypes = [c_float_p, ctypes.c_float, ctypes.c_float, ctypes.c_uint32,
                                                       ctypes.c_uint32, c_float_p]
        L.rtamd_host_generate_primary_rays.restype = None
        _host = L
    return _host


ABI_VERSION = 8   # include/rtamd.h RTAMD_ABI_VERSION


def device_lib():
    global _device
    if _device is None:
        if not os.path.exists(DEVICE_LIB_PATH):
            raise RuntimeError(f"{DEVICE_LIB_PATH} is missing: the HIP library must be built "
                               f"(`make -C raytracer_amd/csrc device`); there is no CPU fallback")
        # One HIP runtime per process.  torch ships its own libamdhip64.so.7 and librtamd.so asks for the
        # system one by the same SONAME, so whichever is loaded first serves both — and torch's device
        # initialisation fails ("No HIP GPUs are available") on top of the system copy.  Python callers of this
        # package use torch for device buffers and streams, so let torch bring its runtime in first.
        if os.environ.get("RTAMD_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = ctypes.CDLL(DEVICE_LIB_PATH)
        L.rtamd_abi_version.restype = ctypes.c_uint32
        if L.rtamd_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{DEVICE_LIB_PATH} has ABI version {L.rtamd_abi_version()}, these bindings are for {ABI_VERSION}: rebuild the library")
        vp = ctypes.c_void_p
        L.rtamd_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        L.rtamd_context_create.restype = ctypes.c_int
        L.rtamd_context_destroy.argtypes = [vp]
        L.rtamd_context_destroy.restype = None
        L.rtamd_last_error.argtypes = [vp]
        L.rtamd_last_error.restype = ctypes.c_char_p
        L.rtamd_context_set_stream.argtypes = [vp, vp]
        L.rtamd_context_set_stream.restype = ctypes.c_int
        L.rtamd_synchronize.argtypes = [vp]
        L.rtamd_synchronize.restype = ctypes.c_int
        L.rtamd_upload_scene.argtypes = [vp, ctypes.POINTER(SceneDesc)]
        L.rtamd_upload_scene.restype = ctypes.c_int
        L.rtamd_upload_texture.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_uint32, ctypes.c_uint32]
        L.rtamd_upload_texture.restype = ctypes.c_int
        L.rtamd_upload_materials.argtypes = [vp, vp, ctypes.c_uint32]
        L.rtamd_upload_materials.restype = ctypes.c_int
        for name in ("rtamd_render", "rtamd_render_device", "rtamd_render_device_static"):
            getattr(L, name).argtypes = [vp, ctypes.POINTER(FrameDesc), ctypes.POINTER(FrameOut)]
            getattr(L, name).restype = ctypes.c_int
        L.rtamd_render_batch_device.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(FrameDesc), ctypes.POINTER(FrameOut)]
        L.rtamd_render_batch_device.restype = ctypes.c_int
        for name in ("rtamd_trace", "rtamd_trace_device"):
            getattr(L, name).argtypes = [vp, ctypes.c_uint64, vp, vp, vp]
            getattr(L, name).restype = ctypes.c_int
        L.rtamd_trace_batch_device.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(TraceJob)]
        L.rtamd_trace_batch_device.restype = ctypes.c_int
        if hasattr(L, "rtamd_trace_image_batch_device"):      # (absent from older builds loaded through RTAMD_DEVICE_LIB for A/B runs)
            L.rtamd_trace_image_batch_device.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(TraceJob), ctypes.c_uint32, ctypes.c_uint32]
            L.rtamd_trace_image_batch_device.restype = ctypes.c_int
        for name in ("rtamd_generate_rays", "rtamd_generate_rays_device"):
            getattr(L, name).argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, c_float_p, ctypes.c_float, ctypes.c_float, vp]
            getattr(L, name).restype = ctypes.c_int
        for name in ("rtamd_surface_from_hits", "rtamd_surface_from_hits_device"):
            getattr(L, name).argtypes = [vp, ctypes.c_uint64, vp, vp, vp, ctypes.c_uint32, vp, vp]
            getattr(L, name).restype = ctypes.c_int
        L.rtamd_generate_rays_offset_device.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, c_float_p, ctypes.c_float, ctypes.c_float,
                                                        ctypes.c_float, ctypes.c_float, vp]
        L.rtamd_generate_rays_offset_device.restype = ctypes.c_int
        L.rtamd_generate_rays_tiles_device.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, c_float_p, ctypes.c_float, ctypes.c_float,
                                                       ctypes.c_float, ctypes.c_float, ctypes.c_uint32, ctypes.c_uint32, vp]
        L.rtamd_generate_rays_tiles_device.restype = ctypes.c_int
        L.rtamd_last_launch_coherence.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32)]
        L.rtamd_last_launch_coherence.restype = ctypes.c_int
        L.rtamd_upload_noise_texture.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint32]
        L.rtamd_upload_noise_texture.restype = ctypes.c_int
        for name in ("rtamd_dispatch_rays", "rtamd_dispatch_rays_device"):
            getattr(L, name).argtypes = [vp, ctypes.POINTER(DispatchDesc), ctypes.POINTER(DispatchOut)]
            getattr(L, name).restype = ctypes.c_int
        L.rtamd_tiles_per_rank.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        L.rtamd_tiles_per_rank.restype = ctypes.c_uint32
        L.rtamd_assemble_tiles_device.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp]
        L.rtamd_assemble_tiles_device.restype = ctypes.c_int
        L.rtamd_assemble_tile_batch_device.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp]
        L.rtamd_assemble_tile_batch_device.restype = ctypes.c_int
        L.rtamd_display_rgba8_device.argtypes = [vp, vp, ctypes.c_uint64, vp]
        L.rtamd_display_rgba8_device.restype = ctypes.c_int
        L.rtamd_get_scene_info.argtypes = [vp, ctypes.POINTER(SceneInfo)]
        L.rtamd_get_scene_info.restype = ctypes.c_int
        L.rtamd_last_kernel_ms.argtypes = [vp, c_float_p]
        L.rtamd_last_kernel_ms.restype = ctypes.c_int
        L.rtamd_set_kernel_variant.argtypes = [vp, ctypes.c_int]
        L.rtamd_set_kernel_variant.restype = ctypes.c_int
        L.rtamd_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64]
        L.rtamd_set_option.restype = ctypes.c_int
        for name in ("rtamd_multibounce_entries", "rtamd_multibounce_primary_rays"):
            getattr(L, name).argtypes = [ctypes.POINTER(MultiBounceDesc)]
            getattr(L, name).restype = ctypes.c_uint64
        L.rtamd_multibounce_prepare_lane_device.argtypes = [vp, ctypes.POINTER(MultiBounceDesc), ctypes.POINTER(MultiBounceLane)]
        L.rtamd_multibounce_prepare_lane_device.restype = ctypes.c_int
        L.rtamd_multibounce_sample_batch_device.argtypes = [vp, ctypes.POINTER(MultiBounceDesc), c_u32_p, ctypes.c_uint32, ctypes.POINTER(MultiBounceLane), vp]
        L.rtamd_multibounce_sample_batch_device.restype = ctypes.c_int
        # one process, several GPUs (rtamd_multi_*)
        L.rtamd_multi_create.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(vp)]
        L.rtamd_multi_create.restype = ctypes.c_int
        L.rtamd_multi_destroy.argtypes = [vp]
        L.rtamd_multi_destroy.restype = None
        L.rtamd_multi_last_error.argtypes = [vp]
        L.rtamd_multi_last_error.restype = ctypes.c_char_p
        L.rtamd_multi_device_count.argtypes = [vp]
        L.rtamd_multi_device_count.restype = ctypes.c_uint32
        L.rtamd_multi_context.argtypes = [vp, ctypes.c_uint32]
        L.rtamd_multi_context.restype = vp
        L.rtamd_multi_upload_scene.argtypes = [vp, ctypes.POINTER(SceneDesc)]
        L.rtamd_multi_upload_scene.restype = ctypes.c_int
        L.rtamd_multi_upload_texture.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_uint32, ctypes.c_uint32]
        L.rtamd_multi_upload_texture.restype = ctypes.c_int
        L.rtamd_multi_upload_materials.argtypes = [vp, vp, ctypes.c_uint32]
        L.rtamd_multi_upload_materials.restype = ctypes.c_int
        L.rtamd_multi_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64]
        L.rtamd_multi_set_option.restype = ctypes.c_int
        for name in ("rtamd_multi_render_batch", "rtamd_multi_render_batch_device"):
            getattr(L, name).argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(FrameDesc), ctypes.c_int, vp]
            getattr(L, name).restype = ctypes.c_int
        L.rtamd_multi_synchronize.argtypes = [vp]
        L.rtamd_multi_synchronize.restype = ctypes.c_int
        L.rtamd_multi_root_stream.argtypes = [vp]
        L.rtamd_multi_root_stream.restype = vp
        L.rtamd_multi_last_gather_ms.argtypes = [vp, c_float_p, c_float_p]
        L.rtamd_multi_last_gather_ms.restype = ctypes.c_int
        L.rtamd_version.argtypes = []
        L.rtamd_version.restype = ctypes.c_char_p
        _device = L
    return _device
