"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

MODE_BVH, MODE_BRUTE, MODE_LEAFBOX = 0, 1, 2

# default scene of the reference's main(): src/main.rs:337-358 and :102-111
EYE = (0.0, 100.0, 200.0)
LOOK_AT = (0.0, 0.0, -100000.0)
UP = (0.0, 1.0, 0.0)
DISTANCE = 288.0
LIGHT_TRI = (-10.0, 300.0, -10.0, 10.0, 300.0, -10.0, 0.0, 300.0, 0.0)
GROUND_TRI = (-10000.0, 0.0, -10000.0, 10000.0, 0.0, -10000.0, 0.0, 0.0, 10000.0)
GROUND_RGB = (0.5, 0.5, 0.5)
NB_RAY, NB_LIGHT_SAMPLE, NB_RAND_SAMPLE = 1, 100, 2000000
SEED = 20261004


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "primary_rays", "primary_hits", "mesh_hits", "shadow_rays", "slab_tests",
        "tri_tests", "assert_tmin_gt_tmax", "nonfinite_t", "exact_ties")] + [
        ("render_ms", C.c_double), ("bvh_build_ms", C.c_double)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Hit(C.Structure):
    _fields_ = [("hit", C.c_int), ("tri", C.c_uint32), ("t", C.c_float), ("p_hit", C.c_float * 3)]


_lib = None
f32p = C.POINTER(C.c_float)


def _fp(a):
    return a.ctypes.data_as(f32p)


def f3(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(3))


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src_m = max(os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("oracle.c", "oracle.h"))
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < src_m:
        build()
    L = C.CDLL(LIB_PATH)
    L.orc_scene_create.restype = C.c_void_p
    L.orc_scene_create.argtypes = [C.c_uint32, C.c_uint32, f32p, f32p, f32p, C.c_float, f32p,
                                   C.c_uint32, f32p, f32p, C.c_uint32, C.c_uint32, f32p, C.c_uint32, C.c_int]
    L.orc_scene_create_ex.restype = C.c_void_p
    L.orc_scene_create_ex.argtypes = [C.c_uint32, C.c_uint32, f32p, f32p, f32p, C.c_float, f32p,
                                      C.c_uint32, f32p, f32p, C.c_uint32, f32p, f32p, C.c_void_p,
                                      C.c_uint32, C.c_uint32, f32p, C.c_uint32, C.c_int]
    L.orc_sphere_intersect.argtypes = [f32p, C.c_float, f32p, f32p, f32p]
    L.orc_sphere_intersect.restype = C.c_int
    L.orc_scene_destroy.argtypes = [C.c_void_p]
    L.orc_bvh_node_count.argtypes = [C.c_void_p]
    L.orc_bvh_node_count.restype = C.c_uint32
    L.orc_bvh_depth.argtypes = [C.c_void_p]
    L.orc_bvh_depth.restype = C.c_uint32
    L.orc_bvh_leaf_order.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.orc_closest_hit.argtypes = [C.c_void_p, C.c_int, f32p, f32p, C.POINTER(Hit)]
    L.orc_render_pixel.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, f32p]
    L.orc_render_rows_ex.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.orc_render_rows_ex.restype = C.c_int
    L.orc_render_window.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                    C.c_void_p, C.POINTER(Stats)]
    L.orc_render_window.restype = C.c_int
    L.orc_gen_samples.argtypes = [C.c_uint64, C.c_uint32, f32p]
    L.orc_import_obj.argtypes = [C.c_char_p, C.POINTER(f32p)]
    L.orc_import_obj.restype = C.c_int
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_camera_new.argtypes = [f32p] * 6
    L.orc_triangle_new.argtypes = [f32p] * 6
    L.orc_triangle_intersect.argtypes = [f32p] * 5 + [f32p]
    L.orc_triangle_intersect.restype = C.c_int
    L.orc_bbox_intersect.argtypes = [f32p] * 4 + [f32p, C.POINTER(C.c_int)]
    L.orc_bbox_intersect.restype = C.c_int
    L.orc_triangle_bbox.argtypes = [f32p] * 5
    L.orc_triangle_get_sample.argtypes = [f32p, f32p, f32p, C.c_float, C.c_float, f32p]
    L.orc_ray_new.argtypes = [f32p, f32p]
    L.orc_color_to_rgb8.argtypes = [f32p, C.POINTER(C.c_uint8)]
    L.orc_create_ray.argtypes = [C.c_uint32] * 5 + [f32p] * 4 + [C.c_float, f32p, C.c_uint32, f32p, f32p]
    _lib = L
    return L


# ---------------------------------------------------------------- inputs

def gen_samples(seed=SEED, n_pairs=NB_RAND_SAMPLE):
    out = np.empty((n_pairs, 2), dtype=np.float32)
    lib().orc_gen_samples(seed, n_pairs, _fp(out))
    return out


def const_samples(value=0.5, n_pairs=NB_RAND_SAMPLE):
    return np.full((n_pairs, 2), value, dtype=np.float32)


def import_obj(path):
    p = f32p()
    n = lib().orc_import_obj(os.fsencode(path), C.byref(p))
    if n < 0:
        raise IOError("orc_import_obj(%s) -> %d" % (path, n))
    arr = np.ctypeslib.as_array(p, shape=(n, 9)).copy()
    lib().orc_free(p)
    return arr


def model_path(name):
    return os.path.join(ROOT, "models", name)


def default_primitives(obj_names):
    """Primitive list of the reference's main(): OBJ meshes first, ground LAST
    (src/main.rs:327-335).  Returns (tris[n,9], rgb[n,3])."""
    parts, cols = [], []
    for nme in obj_names:
        t = import_obj(model_path(nme))
        parts.append(t)
        cols.append(np.ones((len(t), 3), dtype=np.float32))
    parts.append(np.asarray(GROUND_TRI, dtype=np.float32).reshape(1, 9))
    cols.append(np.asarray(GROUND_RGB, dtype=np.float32).reshape(1, 3))
    return np.ascontiguousarray(np.concatenate(parts)), np.ascontiguousarray(np.concatenate(cols))


class Scene:
    def __init__(self, width, height, tris, rgb, samples, *, eye=EYE, look_at=LOOK_AT, up=UP,
                 distance=DISTANCE, light_tri=LIGHT_TRI, nb_ray=NB_RAY,
                 nb_light_sample=NB_LIGHT_SAMPLE, build_bvh=True, spheres=None, sphere_rgb=None, kinds=None):
        self.width, self.height = int(width), int(height)
        self.tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
        self.rgb = np.ascontiguousarray(rgb, dtype=np.float32).reshape(-1, 3)
        self.samples = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1, 2)
        self.spheres = np.zeros((0, 4), np.float32) if spheres is None else \
            np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 4)
        self.sphere_rgb = np.ones((len(self.spheres), 3), np.float32) if sphere_rgb is None else \
            np.ascontiguousarray(sphere_rgb, dtype=np.float32).reshape(-1, 3)
        self.kinds = None if kinds is None else np.ascontiguousarray(kinds, dtype=np.uint8)
        self.n_tris = len(self.tris) + len(self.spheres)       # primitives in the Vec (both arms)
        lt = np.ascontiguousarray(np.asarray(light_tri, dtype=np.float32).reshape(9))
        self.h = lib().orc_scene_create_ex(self.width, self.height, _fp(f3(eye)), _fp(f3(look_at)),
                                           _fp(f3(up)), float(distance), _fp(lt), len(self.tris),
                                           _fp(self.tris), _fp(self.rgb), len(self.spheres), _fp(self.spheres),
                                           _fp(self.sphere_rgb),
                                           self.kinds.ctypes.data if self.kinds is not None else None,
                                           nb_ray, nb_light_sample,
                                           _fp(self.samples), len(self.samples), int(build_bvh))
        if not self.h:
            raise RuntimeError("orc_scene_create failed")

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    __del__ = close

    def node_count(self):
        return lib().orc_bvh_node_count(self.h)

    def depth(self):
        return lib().orc_bvh_depth(self.h)

    def leaf_order(self):
        out = np.empty(self.n_tris, dtype=np.uint32)
        lib().orc_bvh_leaf_order(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    def closest_hit(self, o, d, mode=MODE_BVH):
        h = Hit()
        lib().orc_closest_hit(self.h, mode, _fp(f3(o)), _fp(f3(d)), C.byref(h))
        return h

    def render_pixel(self, px, py, mode=MODE_BVH):
        out = np.zeros(3, dtype=np.float32)
        lib().orc_render_pixel(self.h, mode, px, py, _fp(out))
        return out

    def render_rows(self, row0=0, nrows=None, mode=MODE_BVH, nthreads=None, want_tri=False, want_lin=False):
        if nrows is None:
            nrows = self.height - row0
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        img = np.zeros((nrows, self.width, 3), dtype=np.uint8)
        tri = np.zeros((nrows, self.width), dtype=np.uint32) if want_tri else None
        lin = np.zeros((nrows, self.width, 3), dtype=np.float32) if want_lin else None
        st = Stats()
        rc = lib().orc_render_rows_ex(self.h, mode, row0, nrows, nthreads, img.ctypes.data,
                                      tri.ctypes.data if want_tri else None,
                                      lin.ctypes.data if want_lin else None, C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render_rows_ex -> %d" % rc)
        res = [img, st.asdict()]
        if want_tri:
            res.append(tri)
        if want_lin:
            res.append(lin)
        return tuple(res)

    def render_window(self, col0, row0, ncols, nrows, mode=MODE_BVH, nthreads=None):
        """Columns [col0, col0+ncols) x rows [row0, row0+nrows) of the frame -> (uint8 [nrows, ncols, 3], stats)."""
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        img = np.zeros((nrows, ncols, 3), dtype=np.uint8)
        st = Stats()
        rc = lib().orc_render_window(self.h, mode, col0, row0, ncols, nrows, nthreads, img.ctypes.data, C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render_window -> %d" % rc)
        return img, st.asdict()


def default_scene(obj_names, width, height, samples, **kw):
    tris, rgb = default_primitives(obj_names)
    return Scene(width, height, tris, rgb, samples, **kw)
