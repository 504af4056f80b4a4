"""ctypes wrapper of oracle/_build/libptoracle.so -- the CPU restatement of the reference (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this.  Builds the oracle with
`make -C oracle` when the shared object is missing (gcc only, a few seconds).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libptoracle.so")


def build_oracle():
    src = os.path.join(ORACLE_DIR, "oracle.cpp")
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


class OrcTextureDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("odd", C.c_uint32), ("even", C.c_uint32), ("pad", C.c_uint32), ("color", C.c_double * 3),
                ("scale", C.c_double), ("perlin_seed", C.c_uint64)]


class OrcMaterialDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("texture", C.c_uint32), ("param", C.c_double)]


class OrcObjectDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("material", C.c_uint32), ("p", C.c_double * 12)]


class OrcMeshDesc(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("n_positions", C.c_uint64), ("indices", C.POINTER(C.c_uint32)),
                ("n_indices", C.c_uint64), ("normals", C.POINTER(C.c_float)), ("n_normals", C.c_uint64), ("material_id", C.c_int32),
                ("pad", C.c_int32)]


class OrcMtlDesc(C.Structure):
    _fields_ = [("illum", C.c_int32), ("ni", C.c_float), ("kd", C.c_float * 3), ("ns", C.c_float)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        vp = C.c_void_p
        L.orc_scene_named.restype = vp
        L.orc_scene_named.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(OrcMeshDesc), C.c_int,
                                      C.POINTER(OrcMtlDesc), C.c_int]
        L.orc_scene_custom.restype = vp
        L.orc_scene_custom.argtypes = [C.POINTER(OrcObjectDesc), C.c_int, C.POINTER(OrcMaterialDesc), C.c_int,
                                       C.POINTER(OrcTextureDesc), C.c_int, C.POINTER(OrcMeshDesc), C.c_int, C.POINTER(OrcMtlDesc),
                                       C.c_int, C.c_uint64]
        L.orc_scene_free.argtypes = [vp]
        L.orc_scene_error.restype = C.c_char_p
        L.orc_scene_error.argtypes = [vp]
        L.orc_scene_num_prims.argtypes = [vp]
        L.orc_camera_new.argtypes = [C.POINTER(C.c_double)] * 3 + [C.c_double] * 4 + [C.POINTER(C.c_double)]
        L.orc_scene_camera.argtypes = [C.c_char_p, C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.orc_trace_batch.argtypes = [vp, vp, vp, C.c_uint64, C.c_double, C.c_double, vp, vp, vp, vp]
        L.orc_trace_batch_brute.argtypes = [vp, vp, vp, C.c_uint64, C.c_double, C.c_double, vp, vp]
        L.orc_tree_export.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.orc_render_accumulate.argtypes = [vp, C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.c_uint32,
                                            C.c_uint32, C.c_uint32, C.c_uint64, vp, C.POINTER(C.c_uint64), C.c_int]
        L.orc_resolve.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        L.orc_vec3_kat.argtypes = [C.POINTER(C.c_double)]
        L.orc_reflect.argtypes = [C.POINTER(C.c_double)] * 3
        L.orc_refract.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
        L.orc_reflectance.restype = C.c_double
        L.orc_reflectance.argtypes = [C.c_double, C.c_double]
        L.orc_to_u8.restype = C.c_uint8
        L.orc_to_u8.argtypes = [C.c_double]
        L.orc_philox.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_sample_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp]
        L.orc_splitmix.argtypes = [C.c_uint64, C.c_int, vp]
        L.orc_perlin_table.argtypes = [C.c_uint64, vp]
        L.orc_perlin_get.restype = C.c_double
        L.orc_perlin_get.argtypes = [C.c_uint64, C.POINTER(C.c_double)]
        L.orc_texture_value.argtypes = [C.c_int, C.c_double, C.c_uint64, vp, C.c_int, vp]
        L.orc_scatter_probe.argtypes = [C.c_uint32, C.c_double] + [C.POINTER(C.c_double)] * 3 + [C.c_double, C.POINTER(C.c_double),
                                                                                                 C.c_uint64, C.c_uint32, C.c_uint32,
                                                                                                 C.c_uint32, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class OracleScene:
    def __init__(self, handle):
        self._h = handle
        err = lib().orc_scene_error(handle).decode()
        if err:
            lib().orc_scene_free(handle)
            self._h = None
            raise RuntimeError(err)

    @classmethod
    def named(cls, name, layout_seed=1, bvh_seed=1, perlin_seed=1, meshes=(), mtls=()):
        ms = (OrcMeshDesc * max(1, len(meshes)))(*meshes)
        ml = (OrcMtlDesc * max(1, len(mtls)))(*mtls)
        s = cls(lib().orc_scene_named(name.encode(), layout_seed, bvh_seed, perlin_seed, ms, len(meshes), ml, len(mtls)))
        s._keep = (ms, ml, meshes, mtls)
        return s

    @classmethod
    def custom(cls, objects, materials, textures, meshes=(), mtls=(), bvh_seed=1):
        ob = (OrcObjectDesc * max(1, len(objects)))(*objects)
        ma = (OrcMaterialDesc * max(1, len(materials)))(*materials)
        te = (OrcTextureDesc * max(1, len(textures)))(*textures)
        ms = (OrcMeshDesc * max(1, len(meshes)))(*meshes)
        ml = (OrcMtlDesc * max(1, len(mtls)))(*mtls)
        s = cls(lib().orc_scene_custom(ob, len(objects), ma, len(materials), te, len(textures), ms, len(meshes), ml, len(mtls),
                                       bvh_seed))
        s._keep = (ms, ml, meshes, mtls)
        return s

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().orc_scene_free(h)

    @property
    def num_prims(self):
        return lib().orc_scene_num_prims(self._h)

    def trace(self, origins, dirs, t_min=0.001, t_max=float("inf")):
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        n = o.shape[0]
        prim = np.zeros(n, np.int32)
        t = np.zeros(n)
        normal = np.zeros((n, 3))
        front = np.zeros(n, np.uint8)
        lib().orc_trace_batch(self._h, o.ctypes.data, d.ctypes.data, n, t_min, t_max, prim.ctypes.data, t.ctypes.data,
                              normal.ctypes.data, front.ctypes.data)
        return prim, t, normal, front

    def trace_brute(self, origins, dirs, t_min=0.001, t_max=float("inf")):
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        n = o.shape[0]
        prim = np.zeros(n, np.int32)
        t = np.zeros(n)
        lib().orc_trace_batch_brute(self._h, o.ctypes.data, d.ctypes.data, n, t_min, t_max, prim.ctypes.data, t.ctypes.data)
        return prim, t

    def tree(self):
        """(entries, boxes): entries[i] = -1 for a node (boxes rows in order of appearance) or a primitive rank."""
        n = lib().orc_tree_export(self._h, None, 0, None, 0)
        entries = np.zeros(n, np.int32)
        boxes = np.zeros(n * 6)
        lib().orc_tree_export(self._h, entries.ctypes.data, n, boxes.ctypes.data, n * 6)
        n_nodes = int((entries < 0).sum())
        return entries, boxes[: n_nodes * 6].reshape(-1, 6)

    def render_accumulate(self, cam19, width, height, background, sample_begin, sample_count, max_bounces, seed, threads=0):
        cam = (C.c_double * 19)(*[float(x) for x in cam19])
        accum = np.zeros((height, width, 3))
        rays = C.c_uint64()
        lib().orc_render_accumulate(self._h, cam, width, height, d3(background), sample_begin, sample_count, max_bounces, seed,
                                    accum.ctypes.data, C.byref(rays), threads)
        return accum, rays.value


def camera_new(lookfrom, lookat, up, vfov, aspect, aperture, focus):
    out = (C.c_double * 19)()
    lib().orc_camera_new(d3(lookfrom), d3(lookat), d3(up), vfov, aspect, aperture, focus, out)
    return np.array(out)


def scene_camera(name, t, aspect):
    out = (C.c_double * 19)()
    rc = lib().orc_scene_camera(name.encode(), t, aspect, out)
    if rc != 0:
        raise RuntimeError("unknown scene " + name)
    return np.array(out)


def resolve(accum, spp):
    h, w, _ = accum.shape
    a = np.ascontiguousarray(accum, dtype=np.float64)
    out = np.zeros((h, w, 3), np.uint8)
    lib().orc_resolve(a.ctypes.data, w, h, spp, out.ctypes.data)
    return out


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox(c, k, o)
    return list(o)


def sample_draws(seed, pixel, sample, stream, n):
    out = np.zeros(n)
    lib().orc_sample_draws(seed, pixel, sample, stream, n, out.ctypes.data)
    return out


def texture_value(kind, scale, seed, points):
    p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    out = np.zeros_like(p)
    lib().orc_texture_value(kind, scale, seed, p.ctypes.data, p.shape[0], out.ctypes.data)
    return out


def scatter_probe(mat_type, param, albedo, ray_o, ray_d, t, outward_normal, seed=1, pixel=0, sample=0, stream=1):
    out = (C.c_double * 9)()
    ok = lib().orc_scatter_probe(mat_type, param, d3(albedo), d3(ray_o), d3(ray_d), t, d3(outward_normal), seed, pixel, sample,
                                 stream, out)
    o = np.array(out)
    return bool(ok), o[0:3], o[3:6], o[6:9]
