"""HipRayTracer — Python caller of the C-ABI device library (include/rtamd.h).

It plays the role main.rs + gl_wrapper play in the reference (upload the five
scene slices and two albedo textures, src/main.rs:93-102; dispatch the fused
pass, ray_trace_and_shade.frag; or the split trace pass, ray_trace.frag) but
every call goes straight through librtamd.so — there is no Python or CPU compute
path here, and a missing library or GPU raises.
"""
import ctypes

import numpy as np

from . import _ffi
from .host import FAR, NEAR

NO_HIT = 0xFFFFFFFF
TILE_DIM = 32
TILE_PIXELS = TILE_DIM * TILE_DIM
MAX_TEXTURE_SLOTS = 16
# rtamd_shading_material (include/rtamd.h): Ka, Kd, Ks, Ns, albedo texture slot (-1 = none)
SHADING_MATERIAL_DTYPE = np.dtype([("ambient_color", "<f4", (3,)), ("diffuse_color", "<f4", (3,)), ("specular_color", "<f4", (3,)),
                                   ("specular_exp", "<f4"), ("texture_slot", "<i4")])
assert SHADING_MATERIAL_DTYPE.itemsize == 44
# the constants of ray_trace_and_shade.frag:184-187 as a table entry
def reference_material(texture_slot):
    return ((0.5, 0.5, 0.5), (0.8, 0.8, 0.8), (0.4, 0.4, 0.4), 30.0, texture_slot)


class RtamdError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rtamd error {code}: {message}")
        self.code = code


def _ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


class HipRayTracer:
    def __init__(self, device=0):
        self._lib = _ffi.device_lib()
        h = ctypes.c_void_p()
        rc = self._lib.rtamd_context_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise RtamdError(rc, self._lib.rtamd_last_error(None).decode())
        self._h = h
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rtamd_context_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RtamdError(rc, self._lib.rtamd_last_error(self._h).decode())

    # ---- context ----------------------------------------------------------
    def set_stream(self, hip_stream):
        self._check(self._lib.rtamd_context_set_stream(self._h, ctypes.c_void_p(hip_stream or 0)))

    def synchronize(self):
        self._check(self._lib.rtamd_synchronize(self._h))

    def set_kernel_variant(self, variant):
        self._check(self._lib.rtamd_set_kernel_variant(self._h, int(variant)))

    def set_option(self, name, value):
        self._check(self._lib.rtamd_set_option(self._h, name.encode(), int(value)))

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        self._check(self._lib.rtamd_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def last_launch_coherence(self):
        """(wave-uniform share of the node visits, share of lanes active per visit, re-fill threshold used)."""
        u, l, t = ctypes.c_float(), ctypes.c_float(), ctypes.c_uint32()
        self._check(self._lib.rtamd_last_launch_coherence(self._h, ctypes.byref(u), ctypes.byref(l), ctypes.byref(t)))
        return u.value, l.value, t.value

    def scene_info(self):
        info = _ffi.SceneInfo()
        self._check(self._lib.rtamd_get_scene_info(self._h, ctypes.byref(info)))
        return info

    # ---- uploads ----------------------------------------------------------
    def upload_scene(self, scene):
        """scene: dict(nodes[36 B], triangles[16 B], positions, tex_coords|None, normals|None)."""
        nodes = np.ascontiguousarray(scene["nodes"])
        tris = np.ascontiguousarray(scene["triangles"])
        pos = np.ascontiguousarray(scene["positions"], dtype=np.float32)
        tc = scene.get("tex_coords")
        nr = scene.get("normals")
        tc = None if tc is None else np.ascontiguousarray(tc, dtype=np.float32)
        nr = None if nr is None else np.ascontiguousarray(nr, dtype=np.float32)
        if nodes.size and nodes.dtype.itemsize != 36 or tris.size and tris.dtype.itemsize != 16:
            raise ValueError("nodes must be 36-byte and triangles 16-byte records")
        # the C-ABI takes pointers and ONE vertex count (as the reference's SSBOs do): the per-vertex streams must agree with it
        if pos.size % 3 or (tc is not None and tc.size != 2 * (pos.size // 3)) or (nr is not None and nr.size != pos.size):
            raise ValueError("positions must be (V, 3), tex_coords (V, 2) and normals (V, 3) for the same V")
        d = _ffi.SceneDesc(nodes.ctypes.data if nodes.size else None, len(nodes), tris.ctypes.data if tris.size else None, len(tris),
                           pos.ctypes.data if pos.size else None, pos.size // 3,
                           None if tc is None else tc.ctypes.data, None if nr is None else nr.ctypes.data)
        self._check(self._lib.rtamd_upload_scene(self._h, ctypes.byref(d)))

    def upload_texture(self, slot, rgb8):
        t = np.ascontiguousarray(rgb8, dtype=np.uint8)
        if t.ndim != 3 or t.shape[2] != 3:
            raise ValueError("texture must be (h, w, 3) uint8")
        self._check(self._lib.rtamd_upload_texture(self._h, int(slot), ctypes.c_void_p(t.ctypes.data), t.shape[1], t.shape[0]))

    def upload_materials(self, materials):
        """Per-material shading table (rtamd_upload_materials): an (n,) array of SHADING_MATERIAL_DTYPE, or a
        sequence of (ambient_color, diffuse_color, specular_color, specular_exp, texture_slot) tuples;
        None or an empty sequence restores the shader's constants."""
        if materials is None or len(materials) == 0:
            self._check(self._lib.rtamd_upload_materials(self._h, None, 0))
            return
        m = np.ascontiguousarray(np.array([tuple(x) for x in materials], dtype=SHADING_MATERIAL_DTYPE)
                                 if not isinstance(materials, np.ndarray) else materials, dtype=SHADING_MATERIAL_DTYPE)
        self._check(self._lib.rtamd_upload_materials(self._h, ctypes.c_void_p(m.ctypes.data), len(m)))

    # ---- fused render -----------------------------------------------------
    def _frame_desc(self, width, height, dir_ptr, org_ptr, camera, has_tex_coords, has_normals, tile_rank, tile_world):
        d = _ffi.FrameDesc()
        d.width, d.height = int(width), int(height)
        d.dir, d.org = dir_ptr, org_ptr
        if camera is not None:
            inv, pos = camera
            inv = np.asarray(inv, dtype=np.float32).reshape(16)
            for i in range(16):
                d.inv_proj_view[i] = float(inv[i])
            for i in range(3):
                d.cam_pos[i] = float(np.float32(pos[i]))
        d.near_plane, d.far_plane = NEAR, FAR
        d.has_tex_coords, d.has_normals = int(bool(has_tex_coords)), int(bool(has_normals))
        d.tile_rank, d.tile_world = int(tile_rank), int(tile_world)
        return d

    def tiles_per_rank(self, width, height, tile_world):
        return int(self._lib.rtamd_tiles_per_rank(width, height, tile_world))

    def local_tile_count(self, width, height, tile_rank, tile_world):
        n = ((width + TILE_DIM - 1) // TILE_DIM) * ((height + TILE_DIM - 1) // TILE_DIM)
        return (n - tile_rank + tile_world - 1) // tile_world if tile_rank < n else 0

    def render(self, width, height, dir=None, org=None, camera=None, has_tex_coords=False, has_normals=False,
               tile_rank=0, tile_world=1, want=("rgba", "hit_id", "hit_t")):
        """Host-buffer render (rtamd_render).  Either dir+org arrays (h*w*3 float32) or
        camera=(inv_proj_view[16], position[3]) for on-device ray generation.
        Returns a dict of numpy arrays (flat pixel index; see rtamd_frame_out)."""
        if dir is not None:
            dir = np.ascontiguousarray(dir, dtype=np.float32).reshape(-1, 3)
            org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
            if len(dir) != width * height or len(org) != width * height:
                raise ValueError("dir/org must hold width*height rays")
        elif camera is None:
            raise ValueError("give dir+org or camera")
        n_out = width * height if tile_world == 1 else self.local_tile_count(width, height, tile_rank, tile_world) * TILE_PIXELS
        res = {}
        if "rgba" in want:
            res["rgba"] = np.zeros((n_out, 4), np.float32)
        if "hit_id" in want:
            res["hit_id"] = np.zeros(n_out, np.uint32)
        if "hit_t" in want:
            res["hit_t"] = np.zeros(n_out, np.float32)
        if "rgba8" in want:
            res["rgba8"] = np.zeros((n_out,), np.uint32)
        if "counts" in want:
            res["counts"] = np.zeros((n_out, 2), np.uint32)
        d = self._frame_desc(width, height, _ptr(dir), _ptr(org), camera, has_tex_coords, has_normals, tile_rank, tile_world)
        o = _ffi.FrameOut(_ptr(res.get("rgba")), _ptr(res.get("hit_id")), _ptr(res.get("hit_t")), _ptr(res.get("counts")), _ptr(res.get("rgba8")))
        self._check(self._lib.rtamd_render(self._h, ctypes.byref(d), ctypes.byref(o)))
        return res

    def render_device(self, width, height, dir_ptr=0, org_ptr=0, camera=None, has_tex_coords=False, has_normals=False,
                      tile_rank=0, tile_world=1, rgba_ptr=0, hit_id_ptr=0, hit_t_ptr=0, counts_ptr=0, rgba8_ptr=0):
        """Device-pointer render (rtamd_render_device): integers are raw device addresses
        (e.g. torch.Tensor.data_ptr()); asynchronous on the context's stream."""
        d = self._frame_desc(width, height, ctypes.c_void_p(dir_ptr or None) if dir_ptr else None,
                             ctypes.c_void_p(org_ptr) if org_ptr else None, camera, has_tex_coords, has_normals, tile_rank, tile_world)
        o = _ffi.FrameOut(rgba_ptr or None, hit_id_ptr or None, hit_t_ptr or None, counts_ptr or None, rgba8_ptr or None)
        self._check(self._lib.rtamd_render_device(self._h, ctypes.byref(d), ctypes.byref(o)))

    def render_batch_device(self, width, height, frames, has_tex_coords=False, has_normals=False, tile_rank=0, tile_world=1):
        """rtamd_render_batch_device: up to 16 frames (RTAMD_MAX_BATCH_FRAMES) in one launch.  frames: a sequence of dicts with the per-frame
        keyword arguments of render_device (dir_ptr, org_ptr | camera; rgba_ptr, hit_id_ptr, hit_t_ptr, counts_ptr, rgba8_ptr)."""
        n = len(frames)
        descs = (_ffi.FrameDesc * n)()
        outs = (_ffi.FrameOut * n)()
        for k, fr in enumerate(frames):
            dp, op = fr.get("dir_ptr", 0), fr.get("org_ptr", 0)
            descs[k] = self._frame_desc(width, height, ctypes.c_void_p(dp) if dp else None, ctypes.c_void_p(op) if op else None, fr.get("camera"),
                                        has_tex_coords, has_normals, tile_rank, tile_world)
            outs[k] = _ffi.FrameOut(fr.get("rgba_ptr") or None, fr.get("hit_id_ptr") or None, fr.get("hit_t_ptr") or None,
                                    fr.get("counts_ptr") or None, fr.get("rgba8_ptr") or None)
        self._check(self._lib.rtamd_render_batch_device(self._h, n, descs, outs))

    def render_device_static(self, width, height, dir_ptr=0, org_ptr=0, camera=None, has_tex_coords=False, has_normals=False,
                             tile_rank=0, tile_world=1, rgba_ptr=0, hit_id_ptr=0, hit_t_ptr=0, counts_ptr=0, rgba8_ptr=0):
        """rtamd_render_device_static: render_device for callers that cannot allocate (static launch scratch)."""
        d = self._frame_desc(width, height, ctypes.c_void_p(dir_ptr or None) if dir_ptr else None,
                             ctypes.c_void_p(org_ptr) if org_ptr else None, camera, has_tex_coords, has_normals, tile_rank, tile_world)
        o = _ffi.FrameOut(rgba_ptr or None, hit_id_ptr or None, hit_t_ptr or None, counts_ptr or None, rgba8_ptr or None)
        self._check(self._lib.rtamd_render_device_static(self._h, ctypes.byref(d), ctypes.byref(o)))

    # ---- split-pipeline trace --------------------------------------------
    def trace(self, org, dir):
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, dtype=np.float32).reshape(-1, 3)
        if len(org) != len(dir):
            raise ValueError("org and dir must have the same length")
        hits = np.zeros((len(dir), 4), np.float32)
        self._check(self._lib.rtamd_trace(self._h, len(dir), _ptr(org), _ptr(dir), _ptr(hits)))
        return hits

    def trace_device(self, n, org_ptr, dir_ptr, hits_ptr):
        self._check(self._lib.rtamd_trace_device(self._h, int(n), ctypes.c_void_p(org_ptr), ctypes.c_void_p(dir_ptr), ctypes.c_void_p(hits_ptr)))

    def trace_batch_device(self, jobs):
        """rtamd_trace_batch_device: up to 16 ray lists in one launch.  jobs: a sequence of (n, org_ptr, dir_ptr, hits_ptr)."""
        arr = (_ffi.TraceJob * len(jobs))(*[_ffi.TraceJob(int(n), o or None, d or None, h or None) for n, o, d, h in jobs])
        self._check(self._lib.rtamd_trace_batch_device(self._h, len(jobs), arr))

    def trace_image_batch_device(self, jobs, width, tile_world=1):
        """rtamd_trace_image_batch_device: ray lists that are images (scan-line order of `width` pixels per row, or the tile-major
        shard layout for tile_world > 1), traced in 8x8-pixel work items.  jobs as for trace_batch_device."""
        arr = (_ffi.TraceJob * len(jobs))(*[_ffi.TraceJob(int(n), o or None, d or None, h or None) for n, o, d, h in jobs])
        self._check(self._lib.rtamd_trace_image_batch_device(self._h, len(jobs), arr, int(width), int(tile_world)))

    # ---- ray generation / assembly -----------------------------------------
    def generate_rays(self, width, height, inv_proj_view):
        inv = np.ascontiguousarray(inv_proj_view, dtype=np.float32).reshape(16)
        out = np.zeros((height, width, 3), np.float32)
        self._check(self._lib.rtamd_generate_rays(self._h, width, height, inv.ctypes.data_as(_ffi.c_float_p), NEAR, FAR, _ptr(out)))
        return out

    def generate_rays_device(self, width, height, inv_proj_view, dir_ptr):
        inv = np.ascontiguousarray(inv_proj_view, dtype=np.float32).reshape(16)
        self._check(self._lib.rtamd_generate_rays_device(self._h, width, height, inv.ctypes.data_as(_ffi.c_float_p), NEAR, FAR,
                                                         ctypes.c_void_p(dir_ptr)))

    def generate_rays_offset_device(self, width, height, inv_proj_view, offset_x, offset_y, dir_ptr):
        inv = np.ascontiguousarray(inv_proj_view, dtype=np.float32).reshape(16)
        self._check(self._lib.rtamd_generate_rays_offset_device(self._h, width, height, inv.ctypes.data_as(_ffi.c_float_p), NEAR, FAR,
                                                                float(offset_x), float(offset_y), ctypes.c_void_p(dir_ptr)))

    def generate_rays_tiles_device(self, width, height, inv_proj_view, offset_x, offset_y, tile_rank, tile_world, dir_ptr):
        """rtamd_generate_rays_tiles_device: the rays of one rank's tile shard, compact tile-major layout."""
        inv = np.ascontiguousarray(inv_proj_view, dtype=np.float32).reshape(16)
        self._check(self._lib.rtamd_generate_rays_tiles_device(self._h, width, height, inv.ctypes.data_as(_ffi.c_float_p), NEAR, FAR,
                                                               float(offset_x), float(offset_y), int(tile_rank), int(tile_world), ctypes.c_void_p(dir_ptr)))

    # ---- hit records -> G-buffer ---------------------------------------------
    def surface_from_hits(self, org, dir, hits, has_normals=False):
        o = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dir, dtype=np.float32).reshape(-1, 3)
        h = np.ascontiguousarray(hits, dtype=np.float32).reshape(-1, 4)
        pos = np.zeros((len(h), 3), np.float32)
        nm = np.zeros((len(h), 4), np.float32)
        self._check(self._lib.rtamd_surface_from_hits(self._h, len(h), _ptr(o), _ptr(d), _ptr(h), int(bool(has_normals)), _ptr(pos), _ptr(nm)))
        return pos, nm

    def surface_from_hits_device(self, n, org_ptr, dir_ptr, hits_ptr, has_normals, position_ptr, normal_mat_ptr):
        self._check(self._lib.rtamd_surface_from_hits_device(self._h, int(n), ctypes.c_void_p(org_ptr), ctypes.c_void_p(dir_ptr), ctypes.c_void_p(hits_ptr),
                                                             int(bool(has_normals)), ctypes.c_void_p(position_ptr), ctypes.c_void_p(normal_mat_ptr)))

    # ---- secondary rays (ray_dispatcher.frag) -------------------------------
    def upload_noise_texture(self, rgb8):
        t = np.ascontiguousarray(rgb8, dtype=np.uint8)
        if t.ndim != 3 or t.shape[2] != 3:
            raise ValueError("noise texture must be (h, w, 3) uint8")
        self._check(self._lib.rtamd_upload_noise_texture(self._h, _ptr(t), t.shape[1], t.shape[0]))

    def _dispatch_desc(self, width, height, position_ptr, normal_mat_ptr, light_pos, camera_pos, noise_offset_scale, tile_rank=0, tile_world=1):
        d = _ffi.DispatchDesc()
        d.width, d.height = width, height
        d.tile_rank, d.tile_world = int(tile_rank), int(tile_world)
        d.position, d.normal_mat = position_ptr, normal_mat_ptr
        d.light_pos[:] = [float(x) for x in light_pos]
        d.camera_pos[:] = [float(x) for x in camera_pos]
        d.noise_offset_scale[:] = [float(x) for x in noise_offset_scale]
        return d

    def dispatch_rays(self, position, normal_mat, light_pos, camera_pos, noise_offset_scale):
        """(h, w, 3) positions + (h, w, 4) normal/material G-buffer -> dict of four (h, w, 3) arrays."""
        p = np.ascontiguousarray(position, dtype=np.float32)
        nm = np.ascontiguousarray(normal_mat, dtype=np.float32)
        h, w = nm.shape[:2]
        if p.shape != (h, w, 3) or nm.shape != (h, w, 4):
            raise ValueError("position must be (h, w, 3) and normal_mat (h, w, 4)")
        outs = {k: np.zeros((h, w, 3), np.float32) for k in ("org", "shadow_dir", "reflect_dir", "ambient_dir")}
        d = self._dispatch_desc(w, h, _ptr(p).value, _ptr(nm).value, light_pos, camera_pos, noise_offset_scale)
        o = _ffi.DispatchOut(*(_ptr(outs[k]).value for k in ("org", "shadow_dir", "reflect_dir", "ambient_dir")))
        self._check(self._lib.rtamd_dispatch_rays(self._h, ctypes.byref(d), ctypes.byref(o)))
        return outs

    def dispatch_rays_device(self, width, height, position_ptr, normal_mat_ptr, light_pos, camera_pos, noise_offset_scale,
                             org_ptr=0, shadow_dir_ptr=0, reflect_dir_ptr=0, ambient_dir_ptr=0, tile_rank=0, tile_world=1):
        d = self._dispatch_desc(width, height, position_ptr, normal_mat_ptr, light_pos, camera_pos, noise_offset_scale, tile_rank, tile_world)
        o = _ffi.DispatchOut(org_ptr or None, shadow_dir_ptr or None, reflect_dir_ptr or None, ambient_dir_ptr or None)
        self._check(self._lib.rtamd_dispatch_rays_device(self._h, ctypes.byref(d), ctypes.byref(o)))

    def assemble_tiles_device(self, gathered_ptr, width, height, tile_world, elems_per_pixel, frame_ptr):
        self._check(self._lib.rtamd_assemble_tiles_device(self._h, ctypes.c_void_p(gathered_ptr), width, height, tile_world,
                                                          elems_per_pixel, ctypes.c_void_p(frame_ptr)))


    def display_rgba8_device(self, rgba_ptr, n_pixels, out_ptr):
        """util/display.frag: RGBA32F -> packed RGBA8 (GL unorm conversion) on the device."""
        self._check(self._lib.rtamd_display_rgba8_device(self._h, ctypes.c_void_p(rgba_ptr), int(n_pixels), ctypes.c_void_p(out_ptr)))

    def assemble_tile_batch_device(self, gathered_ptr, width, height, tile_world, elems_per_pixel, n_frames, frames_ptr):
        """gathered [rank][frame][tiles_per_rank*1024 px] -> frames [frame][height*width px] (one gather per batch of frames)."""
        self._check(self._lib.rtamd_assemble_tile_batch_device(self._h, ctypes.c_void_p(gathered_ptr), width, height, tile_world,
                                                               elems_per_pixel, n_frames, ctypes.c_void_p(frames_ptr)))


GATHER_RGBA8, GATHER_RGBA32F = 0, 1
TRANSPORT_RCCL, TRANSPORT_COPY = 0, 1


class MultiGpuRayTracer:
    """rtamd_multi_* (include/rtamd.h): ONE process driving several GPUs — a context per GPU, the scene replicated, frames cut
    into interleaved 32x32 tiles, one ncclGather per batch inside the library, the scan-line frames assembled on devices[0]."""

    def __init__(self, devices, transport=TRANSPORT_RCCL):
        self._lib = _ffi.device_lib()
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        rc = self._lib.rtamd_multi_create(devs, len(devices), int(transport), ctypes.byref(h))
        if rc != 0:
            raise RtamdError(rc, self._lib.rtamd_multi_last_error(None).decode())
        self._h = h
        self.n = len(devices)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rtamd_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RtamdError(rc, self._lib.rtamd_multi_last_error(self._h).decode())

    def upload_scene(self, scene):
        nodes = np.ascontiguousarray(scene["nodes"])
        tris = np.ascontiguousarray(scene["triangles"])
        pos = np.ascontiguousarray(scene["positions"], dtype=np.float32)
        tc, nr = scene.get("tex_coords"), scene.get("normals")
        tc = None if tc is None else np.ascontiguousarray(tc, dtype=np.float32)
        nr = None if nr is None else np.ascontiguousarray(nr, dtype=np.float32)
        d = _ffi.SceneDesc(nodes.ctypes.data, len(nodes), tris.ctypes.data, len(tris), pos.ctypes.data, pos.size // 3,
                           None if tc is None else tc.ctypes.data, None if nr is None else nr.ctypes.data)
        self._check(self._lib.rtamd_multi_upload_scene(self._h, ctypes.byref(d)))

    def upload_texture(self, slot, rgb8):
        t = np.ascontiguousarray(rgb8, dtype=np.uint8)
        self._check(self._lib.rtamd_multi_upload_texture(self._h, int(slot), ctypes.c_void_p(t.ctypes.data), t.shape[1], t.shape[0]))

    def set_option(self, name, value):
        self._check(self._lib.rtamd_multi_set_option(self._h, name.encode(), int(value)))

    def _descs(self, width, height, frames, has_tex_coords, has_normals):
        descs = (_ffi.FrameDesc * len(frames))()
        keep = []
        for k, fr in enumerate(frames):
            d = descs[k]
            d.width, d.height = int(width), int(height)
            if "dir" in fr:     # host rays
                a = np.ascontiguousarray(fr["dir"], dtype=np.float32).reshape(-1, 3)
                o = np.ascontiguousarray(fr["org"], dtype=np.float32).reshape(-1, 3)
                if len(a) != width * height or len(o) != width * height:
                    raise ValueError("dir/org must hold width*height rays")
                keep += [a, o]
                d.dir, d.org = a.ctypes.data, o.ctypes.data
            else:
                inv, pos = fr["camera"]
                inv = np.asarray(inv, dtype=np.float32).reshape(16)
                for i in range(16):
                    d.inv_proj_view[i] = float(inv[i])
                for i in range(3):
                    d.cam_pos[i] = float(np.float32(pos[i]))
            d.near_plane, d.far_plane = NEAR, FAR
            d.has_tex_coords, d.has_normals = int(bool(has_tex_coords)), int(bool(has_normals))
            d.tile_rank, d.tile_world = 0, 1
        return descs, keep

    def render_batch(self, width, height, frames, gather_format=GATHER_RGBA8, has_tex_coords=False, has_normals=False):
        """frames: dicts with "camera": (inv_proj_view, position) or "dir" / "org" host arrays.  Returns (n_frames, h*w) uint32
        for RGBA8, (n_frames, h*w, 4) float32 for RGBA32F (host arrays; synchronous)."""
        descs, keep = self._descs(width, height, frames, has_tex_coords, has_normals)
        n = len(frames)
        out = np.zeros((n, width * height), np.uint32) if gather_format == GATHER_RGBA8 else np.zeros((n, width * height, 4), np.float32)
        self._check(self._lib.rtamd_multi_render_batch(self._h, n, descs, int(gather_format), ctypes.c_void_p(out.ctypes.data)))
        return out

    def render_batch_device(self, width, height, frames, frames_out_ptr, gather_format=GATHER_RGBA8, has_tex_coords=False, has_normals=False):
        """Asynchronous: frames_out_ptr is a device address on devices[0]; see rtamd_multi_render_batch_device."""
        descs, keep = self._descs(width, height, frames, has_tex_coords, has_normals)
        self._check(self._lib.rtamd_multi_render_batch_device(self._h, len(frames), descs, int(gather_format), ctypes.c_void_p(frames_out_ptr)))

    def synchronize(self):
        self._check(self._lib.rtamd_multi_synchronize(self._h))

    def root_stream(self):
        return self._lib.rtamd_multi_root_stream(self._h)

    def last_gather_ms(self):
        g, a = ctypes.c_float(), ctypes.c_float()
        self._check(self._lib.rtamd_multi_last_gather_ms(self._h, ctypes.byref(g), ctypes.byref(a)))
        return g.value, a.value


def assemble_tiles_host(blocks, width, height, tile_world, elems):
    """Pure-numpy statement of the tile layout (used by CPU tests of the sharding
    logic): blocks[r] is rank r's compact tile-major array (n_local*1024, elems)."""
    tiles_x = (width + TILE_DIM - 1) // TILE_DIM
    out = np.zeros((height, width, elems), dtype=blocks[0].dtype)
    ys, xs = np.mgrid[0:height, 0:width]
    g = (ys // TILE_DIM) * tiles_x + xs // TILE_DIM
    r = g % tile_world
    j = g // tile_world
    off = j * TILE_PIXELS + (ys % TILE_DIM) * TILE_DIM + xs % TILE_DIM
    for rank in range(tile_world):
        m = r == rank
        out[m] = np.asarray(blocks[rank]).reshape(-1, elems)[off[m]]
    return out
