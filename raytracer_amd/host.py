"""Python view of the C++ host scene layer (librtamd_host.so).

Names follow the reference: ResourceParser.parse_model
(src/resource/resource_parser.rs:13-47), Model (src/rendering/model.rs:6-42),
Model.build_bvh -> BVHBuilder (src/raytracing/bvh.rs:60-243), Camera
(src/rendering/camera.rs:9-97).  This module only marshals arrays; every
computation happens in raytracer_amd/csrc/host/*.cpp.
"""
import ctypes
import os
import math

import numpy as np

from . import _ffi

NODE_DTYPE = np.dtype([("bounds", "<f4", (6,)), ("is_leaf", "<u4"), ("a", "<u4"), ("b", "<u4")])  # 36 B, types.rs:101-109
TRIANGLE_DTYPE = np.dtype([("p0", "<u4"), ("p1", "<u4"), ("p2", "<u4"), ("mat_idx", "<u4")])      # 16 B, types.rs:4-9
assert NODE_DTYPE.itemsize == 36 and TRIANGLE_DTYPE.itemsize == 16

NEAR = 0.01    # camera.rs:6
FAR = 1000.0   # camera.rs:7


class ResourceParseError(Exception):
    """util/error.rs:17-33 (ResourceParseError wrapped with its line number)."""


def _borrow(ptr, count, dtype):
    if not ptr or count == 0:
        return None
    nbytes = count * np.dtype(dtype).itemsize
    buf = (ctypes.c_char * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


class Model:
    """Owns a native model handle; arrays are copied out on access."""

    def __init__(self, handle):
        self._h = ctypes.c_void_p(handle)
        self._lib = _ffi.host_lib()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.rtamd_model_free(h)
            self._h = None

    @classmethod
    def from_arrays(cls, triangles, positions, tex_coords=None, normals=None):
        tri = np.ascontiguousarray(triangles)
        if tri.dtype != TRIANGLE_DTYPE:
            tri = np.ascontiguousarray(tri, dtype=np.uint32).reshape(-1, 4).view(TRIANGLE_DTYPE).reshape(-1)
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        tc = None if tex_coords is None else np.ascontiguousarray(tex_coords, dtype=np.float32).reshape(-1, 2)
        nr = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        if tc is not None and len(tc) != len(pos) or nr is not None and len(nr) != len(pos):
            raise ValueError("attribute streams must have one entry per vertex")
        out = ctypes.c_void_p()
        rc = _ffi.host_lib().rtamd_model_from_arrays(
            tri.ctypes.data, len(tri), pos.ctypes.data, len(pos),
            None if tc is None else tc.ctypes.data, None if nr is None else nr.ctypes.data, ctypes.byref(out))
        if rc != 0:
            raise ValueError(f"rtamd_model_from_arrays failed ({rc}): triangle index out of range?")
        return cls(out.value)

    def info(self):
        i = _ffi.ModelInfo()
        self._lib.rtamd_model_get_info(self._h, ctypes.byref(i))
        return i

    def build_bvh(self, threads=1):
        """threads > 1: BVHBuilder::build_parallel — the same tree bit for bit, built on several host threads."""
        rc = self._lib.rtamd_model_build_bvh(self._h) if threads <= 1 else self._lib.rtamd_model_build_bvh_parallel(self._h, int(threads))
        if rc != 0:
            raise RuntimeError(f"build_bvh failed ({rc})")

    def has_tex_coords(self):
        return bool(self.info().has_tex_coords)

    def has_normals(self):
        return bool(self.info().has_normals)

    def triangles(self):
        return _borrow(self._lib.rtamd_model_triangles(self._h), self.info().triangle_count, TRIANGLE_DTYPE)

    def positions(self):
        a = _borrow(self._lib.rtamd_model_positions(self._h), self.info().vertex_count * 3, np.float32)
        return None if a is None else a.reshape(-1, 3)

    def tex_coords(self):
        a = _borrow(self._lib.rtamd_model_tex_coords(self._h), self.info().vertex_count * 2, np.float32)
        return None if a is None else a.reshape(-1, 2)

    def normals(self):
        a = _borrow(self._lib.rtamd_model_normals(self._h), self.info().vertex_count * 3, np.float32)
        return None if a is None else a.reshape(-1, 3)

    def bvh_nodes(self):
        """BVH::data() (bvh.rs:49-51) as a structured array, or None before build_bvh()."""
        return _borrow(self._lib.rtamd_model_bvh_nodes(self._h), self.info().node_count, NODE_DTYPE)

    def materials(self):
        return [self._lib.rtamd_model_material_name(self._h, i).decode() for i in range(self.info().material_count)]

    def material_libs(self):
        out, i = [], 0
        while True:
            n = self._lib.rtamd_model_material_lib_name(self._h, i)
            if n is None:
                return out
            out.append(n.decode())
            i += 1

    def scene_arrays(self):
        """The five upload slices of src/main.rs:93-102 as a dict of numpy arrays."""
        nodes = self.bvh_nodes()
        if nodes is None:
            raise RuntimeError("build_bvh() first")
        return {"nodes": nodes, "triangles": self.triangles(), "positions": self.positions(),
                "tex_coords": self.tex_coords(), "normals": self.normals()}


# ---- on-disk cache of a built scene ("BVH built once ... and cached") -------------
# One .npz with the five upload slices exactly as rtamd_upload_scene takes them (36-B nodes, 16-B
# triangles in leaf order, packed attribute streams) plus a format tag; loading needs no rebuild and
# no model object.  np.load with allow_pickle=False: the file carries plain arrays only.
SCENE_CACHE_FORMAT = "rtamd-scene-1"


def save_scene(path, scene):
    arrays = {"format": np.array(SCENE_CACHE_FORMAT), "nodes": np.ascontiguousarray(scene["nodes"]),
              "triangles": np.ascontiguousarray(scene["triangles"]), "positions": np.ascontiguousarray(scene["positions"])}
    for k in ("tex_coords", "normals"):
        if scene.get(k) is not None:
            arrays[k] = np.ascontiguousarray(scene[k])
    tmp = f"{path}.tmp.{os.getpid()}.npz"
    np.savez(tmp, **arrays)
    os.replace(tmp, path)


def load_scene(path):
    z = np.load(path, allow_pickle=False)
    if "format" not in z.files or str(z["format"]) != SCENE_CACHE_FORMAT:
        raise ValueError(f"{path}: not a {SCENE_CACHE_FORMAT} file")
    nodes, tris = z["nodes"], z["triangles"]
    if nodes.dtype != NODE_DTYPE or tris.dtype != TRIANGLE_DTYPE:
        raise ValueError(f"{path}: unexpected record layout")
    return {"nodes": nodes, "triangles": tris, "positions": z["positions"],
            "tex_coords": z["tex_coords"] if "tex_coords" in z.files else None,
            "normals": z["normals"] if "normals" in z.files else None}


class ResourceParser:
    @staticmethod
    def parse_model(data):
        if isinstance(data, str):
            data = data.encode("utf-8")
        out = ctypes.c_void_p()
        err = ctypes.create_string_buffer(1024)
        rc = _ffi.host_lib().rtamd_model_parse_obj(data, len(data), ctypes.byref(out), err, len(err))
        if rc != 0:
            raise ResourceParseError(err.value.decode(errors="replace"))
        return Model(out.value)


    @staticmethod
    def parse_material_lib(data):
        """resource_parser.rs:49-109 -> list of (name, dict) in file order; dict keys follow material.rs:5-21,
        texture entries are None when the Option is None."""
        if isinstance(data, str):
            data = data.encode("utf-8")
        lib = _ffi.host_lib()
        out = ctypes.c_void_p()
        err = ctypes.create_string_buffer(1024)
        rc = lib.rtamd_material_lib_parse(data, len(data), ctypes.byref(out), err, len(err))
        if rc != 0:
            raise ResourceParseError(err.value.decode(errors="replace"))
        try:
            mats = []
            for i in range(lib.rtamd_material_lib_count(out)):
                r = _ffi.MaterialRec()
                lib.rtamd_material_lib_get(out, i, ctypes.byref(r))
                d = {k: tuple(getattr(r, k)) for k in ("ambient_color", "diffuse_color", "specular_color", "transmission_color")}
                d.update({k: getattr(r, k) for k in ("specular_exp", "transmission", "optical_density")})
                d.update({k: (getattr(r, k).decode() if getattr(r, k) is not None else None)
                          for k in ("ambient_tex", "diffuse_tex", "specular_tex", "specular_exp_tex")})
                d["texture_names"] = [d[k] for k in ("ambient_tex", "diffuse_tex", "specular_tex", "specular_exp_tex") if d[k] is not None]
                mats.append((r.name.decode(), d))
            return mats
        finally:
            lib.rtamd_material_lib_free(out)


class Camera:
    """camera.rs:9-97 without the window: the aspect ratio is given directly."""

    def __init__(self, aspect, fov=math.radians(120.0), position=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), direction=(1.0, 0.0, 0.0)):
        self.aspect = float(aspect)
        self.fov = float(np.float32(fov))
        self.position = np.asarray(position, dtype=np.float32)
        self.up = np.asarray(up, dtype=np.float32)
        self.direction = np.asarray(direction, dtype=np.float32)

    @classmethod
    def new_default(cls, aspect):
        return cls(aspect)

    def set_rotation(self, yaw, pitch):
        d = np.zeros(3, np.float32)
        _ffi.host_lib().rtamd_camera_direction(yaw, pitch, d.ctypes.data_as(_ffi.c_float_p))
        self.direction = d

    def add_position(self, x, y, z):
        self.position = (self.position + np.asarray([x, y, z], np.float32)).astype(np.float32)

    def view_proj_matrices(self):
        view = np.zeros(16, np.float32)
        proj = np.zeros(16, np.float32)
        inv = np.zeros(16, np.float32)
        p = lambda a: a.ctypes.data_as(_ffi.c_float_p)
        rc = _ffi.host_lib().rtamd_camera_matrices(self.aspect, self.fov, p(self.position), p(self.up), p(self.direction),
                                                   p(view), p(proj), p(inv))
        if rc != 0:
            raise ValueError("proj*view is singular")
        return view, proj, inv

    def inv_proj_view(self):
        return self.view_proj_matrices()[2]

    def primary_rays(self, width, height):
        """(dir, org) float32 arrays of shape (height, width, 3); row 0 = bottom row."""
        inv = self.inv_proj_view()
        d = np.zeros((height, width, 3), np.float32)
        _ffi.host_lib().rtamd_host_generate_primary_rays(inv.ctypes.data_as(_ffi.c_float_p), NEAR, FAR, width, height,
                                                         d.ctypes.data_as(_ffi.c_float_p))
        o = np.broadcast_to(self.position, d.shape).astype(np.float32).copy()
        return d, o
