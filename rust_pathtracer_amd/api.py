"""ctypes bindings of include/ptgpu.h (device library) and include/pthost.h (host layer).

Everything here is a direct call through the C ABI.  There is deliberately no fallback: a missing library
or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.environ.get("PT_LIBDIR") or os.path.join(_HERE, "lib")  # PT_LIBDIR: A/B builds of the same libraries


class PtError(RuntimeError):
    """A call into the native libraries failed; the message is pt_last_error() / pth_last_error()."""


def _load(name: str) -> C.CDLL:
    path = os.path.join(_LIBDIR, name)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the HIP backend first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C {_HERE}).  There is no CPU fallback."
        )
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


libptgpu = _load("libptgpu.so")
libpthost = _load("libpthost.so")

# ------------------------------------------------------------------ PODs of ptgpu.h
PT_REF_LEAF = 0x80000000
PT_REF_GROUP6 = 0x40000000
PT_REF_INDEX_MASK = 0x3FFFFFFF
PRIM_SPHERE, PRIM_RECT_XY, PRIM_RECT_XZ, PRIM_RECT_YZ, PRIM_TRIANGLE = range(5)
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT = range(4)
TEX_SOLID, TEX_CHECKER, TEX_PERLIN = range(3)


class PtPrim(C.Structure):
    _fields_ = [("type", C.c_uint32), ("material", C.c_uint32), ("d", C.c_double * 12)]


class PtNode(C.Structure):
    _fields_ = [("bmin", C.c_double * 3), ("bmax", C.c_double * 3), ("left", C.c_uint32), ("right", C.c_uint32)]


class PtMaterial(C.Structure):
    _fields_ = [("type", C.c_uint32), ("texture", C.c_uint32), ("param", C.c_double)]


class PtTexture(C.Structure):
    _fields_ = [
        ("type", C.c_uint32),
        ("odd", C.c_uint32),
        ("even", C.c_uint32),
        ("perlin_table", C.c_uint32),
        ("color", C.c_double * 3),
        ("scale", C.c_double),
    ]


class PtSceneDesc(C.Structure):
    _fields_ = [
        ("nodes", C.POINTER(PtNode)),
        ("n_nodes", C.c_uint32),
        ("root", C.c_uint32),
        ("prims", C.POINTER(PtPrim)),
        ("n_prims", C.c_uint32),
        ("materials", C.POINTER(PtMaterial)),
        ("n_materials", C.c_uint32),
        ("textures", C.POINTER(PtTexture)),
        ("n_textures", C.c_uint32),
        ("perlin_tables", C.POINTER(C.c_uint8)),
        ("n_perlin_tables", C.c_uint32),
    ]


class PtCamera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3),
        ("lower_left_corner", C.c_double * 3),
        ("horizontal", C.c_double * 3),
        ("vertical", C.c_double * 3),
        ("u", C.c_double * 3),
        ("v", C.c_double * 3),
        ("lens_radius", C.c_double),
    ]

    def as_array(self) -> np.ndarray:
        """The 19 doubles in declaration order."""
        return np.frombuffer(bytes(self), dtype=np.float64).copy()


class PtHitOut(C.Structure):
    _fields_ = [("prim", C.c_int32), ("front_face", C.c_uint32), ("t", C.c_double), ("normal", C.c_double * 3)]


HIT_DTYPE = np.dtype([("prim", "<i4"), ("front_face", "<u4"), ("t", "<f8"), ("normal", "<f8", (3,))], align=True)
assert HIT_DTYPE.itemsize == C.sizeof(PtHitOut)


class PtStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("paths", C.c_uint64),
        ("passes", C.c_uint32),
        ("launches_trace", C.c_uint32),
        ("launches_shade", C.c_uint32),
        ("launches_path", C.c_uint32),
        ("ms_total", C.c_double),
        ("ms_raygen", C.c_double),
        ("ms_trace", C.c_double),
        ("ms_shade", C.c_double),
        ("ms_fold", C.c_double),
        ("ms_repack", C.c_double),
        ("ms_path", C.c_double),
        ("ms_busy_trace", C.c_double),
        ("ms_busy_shade", C.c_double),
        ("ms_busy_path", C.c_double),
        ("ms_busy_all", C.c_double),
        ("rays_per_bounce", C.c_uint64 * 64),
        ("tail_from", C.c_uint32),
        ("reserved_", C.c_uint32),
    ]


# ------------------------------------------------------------------ PODs of pthost.h
class PthMeshDesc(C.Structure):
    _fields_ = [
        ("positions", C.POINTER(C.c_float)),
        ("n_positions", C.c_uint64),
        ("indices", C.POINTER(C.c_uint32)),
        ("n_indices", C.c_uint64),
        ("normals", C.POINTER(C.c_float)),
        ("n_normals", C.c_uint64),
        ("material_id", C.c_int32),
        ("pad", C.c_int32),
    ]


class PthMtlDesc(C.Structure):
    _fields_ = [("illum", C.c_int32), ("ni", C.c_float), ("kd", C.c_float * 3), ("ns", C.c_float)]


class PthTextureDesc(C.Structure):
    _fields_ = [
        ("type", C.c_uint32),
        ("odd", C.c_uint32),
        ("even", C.c_uint32),
        ("pad", C.c_uint32),
        ("color", C.c_double * 3),
        ("scale", C.c_double),
        ("perlin_seed", C.c_uint64),
    ]


class PthObjectDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("material", C.c_uint32), ("p", C.c_double * 12)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_void_p)

# ------------------------------------------------------------------ prototypes: every symbol the headers declare
_vp = C.c_void_p
_PTGPU_PROTOS = {
    "pt_device_count": (C.c_int, []),
    "pt_context_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "pt_context_destroy": (None, [_vp]),
    "pt_last_error": (C.c_char_p, [_vp]),
    "pt_context_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "pt_scene_upload": (C.c_int, [_vp, C.POINTER(PtSceneDesc)]),
    "pt_trace_batch": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_double, C.c_double, _vp]),
    "pt_trace_batch_static": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_double, C.c_double, _vp]),
    "pt_accum_reset": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "pt_accum_bind": (C.c_int, [_vp, _vp]),
    "pt_render_accumulate": (
        C.c_int,
        [_vp, C.POINTER(PtCamera), C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
         C.POINTER(PtStats)],
    ),
    "pt_resolve": (C.c_int, [_vp, C.c_uint32, _vp]),
    "pt_accum_download": (C.c_int, [_vp, _vp]),
    "pt_accum_upload": (C.c_int, [_vp, _vp]),
    "pt_accum_device_ptr": (_vp, [_vp]),
    "pt_accum_reduce": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "pt_selftest_arith": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "pt_selftest_division": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "pt_selftest_rng": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "pt_selftest_texture": (C.c_int, [_vp, C.c_uint32, _vp, C.c_size_t, _vp]),
    "pt_selftest_guide_tree": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "pt_selftest_guide_tree_device": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "pt_selftest_split_triangle": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
}
_PTHOST_PROTOS = {
    "pth_last_error": (C.c_char_p, []),
    "pth_scene_create": (_vp, [C.c_char_p]),
    "pth_scene_destroy": (None, [_vp]),
    "pth_scene_set_seeds": (None, [_vp, C.c_uint64, C.c_uint64, C.c_uint64]),
    "pth_scene_set_image": (None, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
    "pth_scene_set_model_path": (C.c_int, [_vp, C.c_char_p]),
    "pth_scene_set_model_synthetic": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_double, C.POINTER(C.c_double), C.c_double]),
    "pth_scene_build": (C.c_int, [_vp]),
    "pth_scene_desc": (C.POINTER(PtSceneDesc), [_vp]),
    "pth_scene_camera": (C.c_int, [_vp, C.c_double, C.POINTER(PtCamera)]),
    "pth_scene_settings": (
        C.c_int,
        [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_double),
         C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    ),
    "pth_scene_mesh_count": (C.c_int, [_vp]),
    "pth_scene_mesh": (C.c_int, [_vp, C.c_int, C.POINTER(PthMeshDesc)]),
    "pth_scene_mtl_count": (C.c_int, [_vp]),
    "pth_scene_mtl": (C.c_int, [_vp, C.c_int, C.POINTER(PthMtlDesc)]),
    "pth_world_create": (
        _vp,
        [C.POINTER(PthObjectDesc), C.c_uint32, C.POINTER(PtMaterial), C.c_uint32, C.POINTER(PthTextureDesc), C.c_uint32,
         C.POINTER(PthMeshDesc), C.c_uint32, C.POINTER(PthMtlDesc), C.c_uint32, C.c_uint64],
    ),
    "pth_camera_new": (
        None,
        [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, C.c_double,
         C.POINTER(PtCamera)],
    ),
    "pth_render": (C.c_int, [_vp, C.c_int, C.c_double, C.c_uint64, _vp, C.POINTER(PtStats)]),
    "pth_write_png": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, _vp]),
    "pth_render_frames": (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "pth_set_progress_callback": (None, [_vp, _vp]),
    "pth_scene_set_rebuild_per_frame": (C.c_int, [_vp, C.c_int]),
    "pth_scene_rebuild_seconds": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pth_obj_probe": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
}
for _name, (_res, _args) in _PTGPU_PROTOS.items():
    _f = getattr(libptgpu, _name)  # AttributeError here = the library does not export what ptgpu.h declares
    _f.restype, _f.argtypes = _res, _args
for _name, (_res, _args) in _PTHOST_PROTOS.items():
    _f = getattr(libpthost, _name)
    _f.restype, _f.argtypes = _res, _args

PTGPU_SYMBOLS = tuple(_PTGPU_PROTOS)
PTHOST_SYMBOLS = tuple(_PTHOST_PROTOS)


def _d3(v) -> C.Array:
    return (C.c_double * 3)(*[float(x) for x in v])


def device_count() -> int:
    n = libptgpu.pt_device_count()
    if n < 0:
        raise PtError(libptgpu.pt_last_error(None).decode())
    return n


def accum_reduce(contexts) -> None:
    """pt_accum_reduce: sums the accumulation buffers of `contexts` (one per GPU) into contexts[0]'s with one RCCL reduce."""
    arr = (_vp * len(contexts))(*[c._h for c in contexts])
    rc = libptgpu.pt_accum_reduce(arr, len(contexts))
    if rc != 0:
        raise PtError(f"pt_accum_reduce: status {rc}: {libptgpu.pt_last_error(contexts[0]._h if contexts else None).decode()}")


def selftest_guide_tree(boxes: np.ndarray):
    """Builds and verifies the ordered walk's guide tree over boxes [n, 6] (lo.xyz, hi.xyz) on the host; returns (nodes, depth)."""
    b = np.ascontiguousarray(boxes, dtype=np.float64)
    assert b.ndim == 2 and b.shape[1] == 6
    n_nodes, depth = C.c_uint32(), C.c_uint32()
    rc = libptgpu.pt_selftest_guide_tree(b.ctypes.data, b.shape[0], C.byref(n_nodes), C.byref(depth))
    if rc != 0:
        raise PtError(f"pt_selftest_guide_tree: status {rc}")
    return n_nodes.value, depth.value


def selftest_split_triangle(corners: np.ndarray, ratio: int = 4) -> np.ndarray:
    """The boxes [k, 6] (lo.xyz, hi.xyz, unpadded) of the pieces by which the guide tree references the triangle `corners` [3, 3]."""
    c = np.ascontiguousarray(corners, dtype=np.float64).reshape(9)
    out = np.zeros((64, 6), dtype=np.float64)
    n = C.c_size_t()
    rc = libptgpu.pt_selftest_split_triangle(c.ctypes.data, ratio, out.ctypes.data, out.shape[0], C.byref(n))
    if rc != 0:
        raise PtError(f"pt_selftest_split_triangle: status {rc}")
    return out[: n.value].copy()


def write_png(path: str, rgb: np.ndarray):
    """write_file (main.rs:78-91): uint8 [h, w, 3] -> 8-bit RGB PNG."""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 3
    if libpthost.pth_write_png(os.fsencode(path), a.shape[1], a.shape[0], a.ctypes.data) != 0:
        raise PtError(libpthost.pth_last_error().decode())


def camera_new(lookfrom, lookat, up, vertical_fov, aspect_ratio, aperture, focus_dist) -> PtCamera:
    """Camera::new (camera.rs:14-44)."""
    cam = PtCamera()
    libpthost.pth_camera_new(_d3(lookfrom), _d3(lookat), _d3(up), vertical_fov, aspect_ratio, aperture, focus_dist, C.byref(cam))
    return cam


class Scene:
    """One of scene.rs's six scenes (or a custom world): builder + flattened tables, host side only."""

    def __init__(self, name: str | None = None, *, _handle=None):
        if _handle is not None:
            self._h = _handle
        else:
            self._h = libpthost.pth_scene_create(name.encode())
            if not self._h:
                raise PtError(libpthost.pth_last_error().decode())
        self.name = name
        self._keep = []

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            libpthost.pth_scene_destroy(h)

    def _ok(self, rc):
        if rc != 0:
            raise PtError(libpthost.pth_last_error().decode())

    def set_seeds(self, layout=1, bvh=1, perlin=1):
        libpthost.pth_scene_set_seeds(self._h, layout, bvh, perlin)
        return self

    def set_image(self, width, height, spp, max_bounces, background):
        libpthost.pth_scene_set_image(self._h, width, height, spp, max_bounces, _d3(background))
        return self

    def set_model_path(self, path: str):
        self._ok(libpthost.pth_scene_set_model_path(self._h, path.encode()))
        return self

    def set_model_synthetic(self, n_lon, n_lat, radius, center, world_scale=1.0):
        self._ok(libpthost.pth_scene_set_model_synthetic(self._h, n_lon, n_lat, radius, _d3(center), world_scale))
        return self

    def build(self):
        self._ok(libpthost.pth_scene_build(self._h))
        return self

    @property
    def desc(self) -> PtSceneDesc:
        p = libpthost.pth_scene_desc(self._h)
        if not p:
            raise PtError(libpthost.pth_last_error().decode())
        return p.contents

    def camera(self, t: float = 0.0) -> PtCamera:
        cam = PtCamera()
        self._ok(libpthost.pth_scene_camera(self._h, t, C.byref(cam)))
        return cam

    def settings(self) -> dict:
        w, h, spp, depth = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        bg = (C.c_double * 3)()
        anim, fps, dur = C.c_int(), C.c_double(), C.c_double()
        self._ok(libpthost.pth_scene_settings(self._h, C.byref(w), C.byref(h), C.byref(spp), C.byref(depth), bg, C.byref(anim),
                                              C.byref(fps), C.byref(dur)))
        return dict(width=w.value, height=h.value, spp=spp.value, max_bounces=depth.value, background=tuple(bg),
                    animation=bool(anim.value), fps=fps.value, duration=dur.value)

    def meshes(self) -> list[PthMeshDesc]:
        out = []
        for i in range(libpthost.pth_scene_mesh_count(self._h)):
            m = PthMeshDesc()
            self._ok(libpthost.pth_scene_mesh(self._h, i, C.byref(m)))
            out.append(m)
        return out

    def mtls(self) -> list[PthMtlDesc]:
        out = []
        for i in range(libpthost.pth_scene_mtl_count(self._h)):
            m = PthMtlDesc()
            self._ok(libpthost.pth_scene_mtl(self._h, i, C.byref(m)))
            out.append(m)
        return out

    def render(self, device: int = 0, t: float = 0.0, seed: int = 0):
        """renderer::render (renderer.rs:5-37) on the GPU: returns (uint8 array [h, w, 3], PtStats)."""
        s = self.settings()
        out = np.empty((s["height"], s["width"], 3), dtype=np.uint8)
        stats = PtStats()
        self._ok(libpthost.pth_render(self._h, device, t, seed, out.ctypes.data, C.byref(stats)))
        return out, stats

    def render_frames(self, out_dir: str, device: int = 0, seed: int = 0, first_frame: int = 0, frame_stride: int = 1, max_frames: int = 0,
                      rebuild_per_frame: bool = False, progress=None):
        """The frame loop of main.rs:24-72: renders the scene's frames (one for a static image) and writes
        out_dir/image_%04d.png, encoding each PNG while the next frame renders.  rebuild_per_frame: BvhNode::new and the
        upload again before every frame (BASELINE.json configs[4]); their time is reported apart."""
        total, written, rays = C.c_uint32(), C.c_uint32(), C.c_uint64()
        t_render, t_write, t_build, t_upload = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        self._ok(libpthost.pth_scene_set_rebuild_per_frame(self._h, 1 if rebuild_per_frame else 0))
        cb = PROGRESS_FN(lambda done, todo, frame, seconds, _user: progress(done, todo, frame, seconds)) if progress else None
        libpthost.pth_set_progress_callback(C.cast(cb, _vp) if cb else None, None)
        try:
            self._render_frames_call(device, seed, out_dir, first_frame, frame_stride, max_frames, total, written, t_render, t_write, rays)
        finally:
            libpthost.pth_set_progress_callback(None, None)
        self._ok(libpthost.pth_scene_rebuild_seconds(self._h, C.byref(t_build), C.byref(t_upload)))
        return {"frames_total": total.value, "frames_written": written.value, "seconds_render": t_render.value,
                "seconds_write": t_write.value, "rays": rays.value, "seconds_build": t_build.value, "seconds_upload": t_upload.value}

    def _render_frames_call(self, device, seed, out_dir, first_frame, frame_stride, max_frames, total, written, t_render, t_write, rays):
        self._ok(libpthost.pth_render_frames(self._h, device, seed, os.fsencode(out_dir), first_frame, frame_stride, max_frames,
                                             C.byref(total), C.byref(written), C.byref(t_render), C.byref(t_write), C.byref(rays)))

    # flattened tables as numpy views (copies), for tests
    def nodes_array(self) -> np.ndarray:
        d = self.desc
        dt = np.dtype([("bmin", "<f8", (3,)), ("bmax", "<f8", (3,)), ("left", "<u4"), ("right", "<u4")])
        return np.frombuffer(C.string_at(d.nodes, d.n_nodes * C.sizeof(PtNode)), dtype=dt).copy()

    def prims_array(self) -> np.ndarray:
        d = self.desc
        dt = np.dtype([("type", "<u4"), ("material", "<u4"), ("d", "<f8", (12,))])
        return np.frombuffer(C.string_at(d.prims, d.n_prims * C.sizeof(PtPrim)), dtype=dt).copy()


def world_create(objects, materials, textures, meshes=(), mtls=(), bvh_seed=1) -> Scene:
    """A world from an object list (Vec<Box<dyn Hittable>> -> BvhNode::new).  Arguments are sequences of the
    ctypes structures PthObjectDesc / PtMaterial / PthTextureDesc / PthMeshDesc / PthMtlDesc."""
    objs = (PthObjectDesc * max(1, len(objects)))(*objects)
    mats = (PtMaterial * max(1, len(materials)))(*materials)
    texs = (PthTextureDesc * max(1, len(textures)))(*textures)
    ms = (PthMeshDesc * max(1, len(meshes)))(*meshes)
    ml = (PthMtlDesc * max(1, len(mtls)))(*mtls)
    h = libpthost.pth_world_create(objs, len(objects), mats, len(materials), texs, len(textures), ms, len(meshes), ml, len(mtls),
                                   bvh_seed)
    if not h:
        raise PtError(libpthost.pth_last_error().decode())
    return Scene(None, _handle=h)


class Context:
    """One GPU's device context (ptgpu.h)."""

    def __init__(self, device: int = 0):
        h = _vp()
        rc = libptgpu.pt_context_create(device, C.byref(h))
        if rc != 0:
            raise PtError(f"pt_context_create({device}) = {rc}: {libptgpu.pt_last_error(None).decode()}")
        self._h = h
        self.device = device
        self._size = None

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            libptgpu.pt_context_destroy(h)

    __del__ = close

    def _ok(self, rc):
        if rc != 0:
            raise PtError(f"status {rc}: {libptgpu.pt_last_error(self._h).decode()}")

    def set_option(self, key: str, value: int):
        self._ok(libptgpu.pt_context_set_option(self._h, key.encode(), value))

    def upload(self, scene: "Scene | PtSceneDesc"):
        d = scene.desc if isinstance(scene, Scene) else scene
        self._ok(libptgpu.pt_scene_upload(self._h, C.byref(d)))

    def trace_batch(self, origins: np.ndarray, dirs: np.ndarray, t_min=0.001, t_max=float("inf")) -> np.ndarray:
        """BvhNode::hit for n rays: structured array with fields prim, front_face, t, normal."""
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        assert o.shape == d.shape
        out = np.zeros(o.shape[0], dtype=HIT_DTYPE)
        self._ok(libptgpu.pt_trace_batch(self._h, o.ctypes.data, d.ctypes.data, o.shape[0], t_min, t_max, out.ctypes.data))
        return out

    def trace_batch_static(self, origins: np.ndarray, dirs: np.ndarray, t_min=0.001, t_max=float("inf")) -> np.ndarray:
        """pt_trace_batch_static: the same query staged in the library's fixed scratch array (no device allocation)."""
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        assert o.shape == d.shape
        out = np.zeros(o.shape[0], dtype=HIT_DTYPE)
        self._ok(libptgpu.pt_trace_batch_static(self._h, o.ctypes.data, d.ctypes.data, o.shape[0], t_min, t_max, out.ctypes.data))
        return out

    def accum_reset(self, width: int, height: int):
        self._ok(libptgpu.pt_accum_reset(self._h, width, height))
        self._size = (width, height)

    def accum_bind(self, device_ptr: int | None):
        self._ok(libptgpu.pt_accum_bind(self._h, device_ptr))

    def accum_device_ptr(self) -> int:
        return libptgpu.pt_accum_device_ptr(self._h)

    def render_accumulate(self, camera: PtCamera, width, height, background, sample_begin, sample_count, max_bounces, seed) -> PtStats:
        stats = PtStats()
        self._ok(libptgpu.pt_render_accumulate(self._h, C.byref(camera), width, height, _d3(background), sample_begin, sample_count,
                                               max_bounces, seed, C.byref(stats)))
        return stats

    def resolve(self, total_spp: int) -> np.ndarray:
        w, h = self._size
        out = np.empty((h, w, 3), dtype=np.uint8)
        self._ok(libptgpu.pt_resolve(self._h, total_spp, out.ctypes.data))
        return out

    def accum_download(self) -> np.ndarray:
        w, h = self._size
        out = np.empty((h, w, 3), dtype=np.float32)
        self._ok(libptgpu.pt_accum_download(self._h, out.ctypes.data))
        return out

    def accum_upload(self, arr: np.ndarray):
        w, h = self._size
        a = np.ascontiguousarray(arr, dtype=np.float32)
        assert a.size == w * h * 3
        self._ok(libptgpu.pt_accum_upload(self._h, a.ctypes.data))

    def selftest_arith(self, a: np.ndarray, b: np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        q, r = np.empty_like(a), np.empty_like(a)
        self._ok(libptgpu.pt_selftest_arith(self._h, a.ctypes.data, b.ctypes.data, a.size, q.ctypes.data, r.ctypes.data))
        return q, r

    def selftest_division(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        q = np.empty_like(a)
        self._ok(libptgpu.pt_selftest_division(self._h, a.ctypes.data, b.ctypes.data, a.size, q.ctypes.data))
        return q

    def selftest_guide_tree_device(self, boxes: np.ndarray):
        """Builds the guide tree over boxes [n, 6] ON THE DEVICE and verifies it on the host; returns (nodes, depth)."""
        b = np.ascontiguousarray(boxes, dtype=np.float64)
        assert b.ndim == 2 and b.shape[1] == 6
        n_nodes, depth = C.c_uint32(), C.c_uint32()
        self._ok(libptgpu.pt_selftest_guide_tree_device(self._h, b.ctypes.data, b.shape[0], C.byref(n_nodes), C.byref(depth)))
        return n_nodes.value, depth.value

    def selftest_rng(self, seed, pixel, sample, stream, n) -> np.ndarray:
        out = np.empty(n, dtype=np.float64)
        self._ok(libptgpu.pt_selftest_rng(self._h, seed, pixel, sample, stream, n, out.ctypes.data))
        return out

    def selftest_texture(self, tex: int, points: np.ndarray) -> np.ndarray:
        p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        out = np.empty_like(p)
        self._ok(libptgpu.pt_selftest_texture(self._h, tex, p.ctypes.data, p.shape[0], out.ctypes.data))
        return out
