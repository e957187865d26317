"""ctypes binding of librtx.so (include/rtx.h) — the Python face of the C ABI.

Plumbing only: every ray is traced by the HIP kernel inside librtx.so.  There is
no Python or CPU rendering path; if the library is missing, importing this
module fails loudly, and rendering without a GPU raises RtxError(NO_DEVICE).

Names mirror the reference (antoinedesbois/Ray-Tracer-Rust): Scene{width,height,
light,camera,bvh} (src/tracer/utils/scene.rs:6-12) is flattened into
RtxSceneDesc; the default-scene literals are those of main() (src/main.rs:327-358).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# librtx.so is the product.  RTX_PY_ABLATION=1 makes THIS BINDING load librtx_ablation.so instead (the same sources
# built with -DRTX_ABLATION=1: every earlier kernel form, selectable through RTX_VARIANT) — for tools/ and the
# variant-equality test only; the library itself reads no environment variable.  RTX_PY_LIB=<path> makes this binding
# load a library built somewhere else (tools/ab_build.sh: A/B variants live under gpurun_out/ and never replace the
# product's file).
LIB_PATH = os.environ.get("RTX_PY_LIB") or os.path.join(
    _HERE, "librtx_ablation.so" if os.environ.get("RTX_PY_ABLATION") == "1" else "librtx.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "%s is not built. Build it with: python -c 'import __graft_entry__ as g; g.build()' "
        "or make -C ray-tracer-rust_amd/csrc [ablation]" % LIB_PATH)

_lib = C.CDLL(LIB_PATH)

OK, ERR_BAD_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_UNSUPPORTED, ERR_INTERNAL, ERR_IO = 0, -1, -2, -3, -4, -5, -6, -7
ACCEL_BVH, ACCEL_BRUTE = 0, 1
REFTREE_AUTO, REFTREE_ALWAYS, REFTREE_NEVER = 0, 1, 2

# constants of the reference: src/main.rs:38-40
NB_RAY, NB_LIGHT_SAMPLE, NB_RAND_SAMPLE = 1, 100, 2000000
# default scene: camera src/main.rs:353-356, light :337-343, ground :102-111, frame :350-351
DEFAULT_EYE = (0.0, 100.0, 200.0)
DEFAULT_LOOK_AT = (0.0, 0.0, -100000.0)
DEFAULT_UP = (0.0, 1.0, 0.0)
DEFAULT_DISTANCE = 288.0
DEFAULT_LIGHT = (-10.0, 300.0, -10.0, 10.0, 300.0, -10.0, 0.0, 300.0, 0.0)
GROUND_TRI = (-10000.0, 0.0, -10000.0, 10000.0, 0.0, -10000.0, 0.0, 0.0, 10000.0)
GROUND_RGB = (0.5, 0.5, 0.5)
MESH_RGB = (1.0, 1.0, 1.0)
DEFAULT_WIDTH, DEFAULT_HEIGHT = 1920, 1080
DEFAULT_SEED = 20261004

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)


class SceneDesc(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("eye", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
        ("distance", C.c_float),
        ("light_v0", C.c_float * 3), ("light_v1", C.c_float * 3), ("light_v2", C.c_float * 3),
        ("n_tris", C.c_uint32),
        ("v0v1v2", f32p), ("rgb", f32p), ("tie_rank", u32p),
        ("nb_ray", C.c_uint32), ("nb_light_sample", C.c_uint32),
        ("samples", f32p), ("n_samples", C.c_uint32),
        ("accel", C.c_uint32), ("leaf_max", C.c_uint32), ("reference_tree", C.c_uint32),
        ("n_spheres", C.c_uint32), ("spheres", f32p), ("sphere_rgb", f32p), ("kinds", u8p),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "primary_rays", "primary_hits", "shadow_rays", "rays", "box_tests", "tri_tests",
        "wave_node_visits", "wave_tri_visits", "redo_tiles")] + [("kernel_ms", C.c_double), ("total_ms", C.c_double)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SceneInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_tris", "n_nodes", "n_leaves", "max_leaf_tris", "depth", "n_light_points",
                                          "n_ref_nodes", "n_global")] + \
               [(n, C.c_uint64) for n in ("node_bytes", "tri_bytes", "shade_bytes", "sample_bytes")]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/rtx.h declares (tests check the export list against the header)
_SIGS = {
    "rtx_abi_version": (C.c_int, []),
    "rtx_device_count": (C.c_int, []),
    "rtx_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]),
    "rtx_scene_destroy": (None, [C.c_void_p]),
    "rtx_scene_info": (C.c_int, [C.c_void_p, C.POINTER(SceneInfo)]),
    "rtx_scene_upload": (C.c_int, [C.c_void_p, C.c_int]),
    "rtx_render_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(Stats)]),
    "rtx_render_frame": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_uint32, C.c_void_p, C.POINTER(Stats)]),
    "rtx_render_tiles_device": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_size_t, C.c_void_p, C.c_void_p]),
    "rtx_tiles_rows": (C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rtx_tiles_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rtx_debug_wave_profile": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, u64p, C.c_size_t, u32p, u32p]),
    "rtx_launch_timings": (C.c_int, [C.c_void_p, C.c_int, C.c_int, f32p, f32p]),
    "rtx_debug_tile_descs": (C.c_int, [C.c_void_p, C.c_int, u32p, C.c_size_t]),
    "rtx_strerror": (C.c_char_p, [C.c_int]),
    "rtx_last_hip_error": (C.c_int, []),
    "rtx_scene_light_points": (C.c_int, [C.c_void_p, f32p]),
    "rtx_scene_gamma_thresholds": (C.c_int, [C.c_void_p, f32p]),
    "rtx_scene_normals": (C.c_int, [C.c_void_p, f32p]),
    "rtx_scene_nodes": (C.c_int, [C.c_void_p, u32p, u32p]),
    "rtx_scene_primary_nodes": (C.c_int, [C.c_void_p, u32p, u32p]),
    "rtx_scene_ref_nodes": (C.c_int, [C.c_void_p, u32p]),
    "rtxh_camera_new": (None, [f32p] * 6),
    "rtxh_import_obj": (C.c_int, [C.c_char_p, C.POINTER(f32p)]),
    "rtxh_import_obj_ex": (C.c_int, [C.c_char_p, C.c_uint32, C.POINTER(f32p), C.POINTER(f32p)]),
    "rtxh_free": (None, [C.c_void_p]),
    "rtxh_ref_leaf_rank": (C.c_int, [C.c_uint32, f32p, u32p]),
    "rtxh_gen_samples": (None, [C.c_uint64, C.c_uint32, f32p]),
    "rtxh_write_png": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "rtxh_synthetic_mesh": (C.c_int, [C.c_uint64, C.c_uint32, f32p]),
    "rtxh_scatter_tiles": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(_lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


class RtxError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = _lib.rtx_strerror(code).decode()
        if code == ERR_HIP:
            msg += " (hipError %d)" % _lib.rtx_last_hip_error()
        super().__init__("%s: %s [%d]" % (where, msg, code))


def _check(code, where):
    if code != OK:
        raise RtxError(code, where)


def _fp(a):
    return a.ctypes.data_as(f32p)


def _f3(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(3))


def abi_version():
    return _lib.rtx_abi_version()


def device_count():
    return _lib.rtx_device_count()


# ------------------------------------------------------------------ host helpers (rtxh_*)
def camera_new(eye, look_at, up):
    """Camera::new (src/tracer/utils/camera.rs:17-35) -> (u, v, w)."""
    u, v, w = (np.zeros(3, np.float32) for _ in range(3))
    _lib.rtxh_camera_new(_fp(_f3(eye)), _fp(_f3(look_at)), _fp(_f3(up)), _fp(u), _fp(v), _fp(w))
    return u, v, w


def import_obj(path):
    """import_obj (src/main.rs:114-149) -> float32 [n, 9] (v0, v1, v2 per triangle)."""
    p = f32p()
    n = _lib.rtxh_import_obj(os.fsencode(path), C.byref(p))
    if n < 0:
        raise RtxError(n, "rtxh_import_obj(%s)" % path)
    arr = np.ctypeslib.as_array(p, shape=(max(n, 1), 9))[:n].copy() if n else np.zeros((0, 9), np.float32)
    _lib.rtxh_free(p)
    return arr


OBJ_SLASHES, OBJ_RELATIVE, OBJ_POLYGONS, OBJ_MATERIALS, OBJ_ALL = 1, 2, 4, 8, 15


def import_obj_ex(path, flags=OBJ_ALL):
    """The loader with slashes, relative indices, polygons and material colours (each opt-in; flags = 0 is
    import_obj with every triangle white) -> (float32 [n, 9], float32 [n, 3])."""
    t, c = f32p(), f32p()
    n = _lib.rtxh_import_obj_ex(os.fsencode(path), flags, C.byref(t), C.byref(c))
    if n < 0:
        raise RtxError(n, "rtxh_import_obj_ex(%s)" % path)
    tris = np.ctypeslib.as_array(t, shape=(max(n, 1), 9))[:n].copy() if n else np.zeros((0, 9), np.float32)
    rgb = np.ctypeslib.as_array(c, shape=(max(n, 1), 3))[:n].copy() if n else np.zeros((0, 3), np.float32)
    _lib.rtxh_free(t)
    _lib.rtxh_free(c)
    return tris, rgb


def ref_leaf_rank(tris):
    tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    out = np.zeros(len(tris), dtype=np.uint32)
    _check(_lib.rtxh_ref_leaf_rank(len(tris), _fp(tris), out.ctypes.data_as(u32p)), "rtxh_ref_leaf_rank")
    return out


def gen_samples(seed=DEFAULT_SEED, n_pairs=NB_RAND_SAMPLE):
    out = np.empty((n_pairs, 2), dtype=np.float32)
    _lib.rtxh_gen_samples(seed, n_pairs, _fp(out))
    return out


def write_png(path, img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    assert c == 3
    _check(_lib.rtxh_write_png(os.fsencode(path), w, h, img.ctypes.data), "rtxh_write_png")


SYNTHETIC_SEED = 12345


def synthetic_mesh(n_tris, seed=SYNTHETIC_SEED):
    """BASELINE.json configs[4]: random triangle soup in the big_bunny AABB -> float32 [n, 9]."""
    out = np.empty((n_tris, 9), dtype=np.float32)
    _check(_lib.rtxh_synthetic_mesh(seed, n_tris, _fp(out)), "rtxh_synthetic_mesh")
    return out


def synthetic_primitives(n_tris, seed=SYNTHETIC_SEED):
    """Synthetic soup (colour 1,1,1) + the ground, ground last as in main() (src/main.rs:335)."""
    t = synthetic_mesh(n_tris, seed)
    tris = np.concatenate([t, np.asarray(GROUND_TRI, np.float32).reshape(1, 9)])
    rgb = np.concatenate([np.tile(np.asarray(MESH_RGB, np.float32), (n_tris, 1)),
                          np.asarray(GROUND_RGB, np.float32).reshape(1, 3)])
    return np.ascontiguousarray(tris), np.ascontiguousarray(rgb)


def default_primitives(obj_paths):
    """The primitive list main() builds: OBJ meshes in argv order, ground last (src/main.rs:327-335)."""
    parts, cols = [], []
    for p in obj_paths:
        t = import_obj(p)
        parts.append(t)
        cols.append(np.tile(np.asarray(MESH_RGB, np.float32), (len(t), 1)))
    parts.append(np.asarray(GROUND_TRI, np.float32).reshape(1, 9))
    cols.append(np.asarray(GROUND_RGB, np.float32).reshape(1, 3))
    return np.ascontiguousarray(np.concatenate(parts)), np.ascontiguousarray(np.concatenate(cols))


# ------------------------------------------------------------------ Scene
class Scene:
    """Flattened Scene{width,height,light,camera,bvh}; owns an RtxScene handle."""

    def __init__(self, width, height, tris, rgb, samples, *, eye=DEFAULT_EYE, look_at=DEFAULT_LOOK_AT,
                 up=DEFAULT_UP, distance=DEFAULT_DISTANCE, light_tri=DEFAULT_LIGHT, nb_ray=NB_RAY,
                 nb_light_sample=NB_LIGHT_SAMPLE, accel=ACCEL_BVH, leaf_max=0, tie_rank="reference",
                 reference_tree=REFTREE_AUTO, spheres=None, sphere_rgb=None, kinds=None):
        """spheres [m, 4] (origin x, y, z, radius) and sphere_rgb [m, 3] are the Sphere arm of Primitive
        (src/tracer/primitives/sphere.rs:12-29); kinds (uint8 per primitive, 0 = next triangle, 1 = next sphere)
        gives the order of the Vec<Primitive>, default all triangles then all spheres.  tie_rank, when given,
        is indexed by position in that Vec."""
        self._h = C.c_void_p()
        self.width, self.height = int(width), int(height)
        self.tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
        self.rgb = np.ascontiguousarray(rgb, dtype=np.float32).reshape(-1, 3)
        self.samples = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1, 2)
        if len(self.rgb) != len(self.tris):
            raise ValueError("rgb and tris disagree")
        self.spheres = np.zeros((0, 4), np.float32) if spheres is None else \
            np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 4)
        self.sphere_rgb = np.ones((len(self.spheres), 3), np.float32) if sphere_rgb is None else \
            np.ascontiguousarray(sphere_rgb, dtype=np.float32).reshape(-1, 3)
        if len(self.sphere_rgb) != len(self.spheres):
            raise ValueError("sphere_rgb and spheres disagree")
        self.kinds = None if kinds is None else np.ascontiguousarray(kinds, dtype=np.uint8).reshape(-1)
        if self.kinds is not None and len(self.kinds) != len(self.tris) + len(self.spheres):
            raise ValueError("kinds must have one entry per primitive")
        self.n_prims = len(self.tris) + len(self.spheres)
        if isinstance(tie_rank, str):
            if tie_rank != "reference":
                raise ValueError(tie_rank)
            rank = None          # the library derives it from the reference tree it builds (reference_tree)
        else:
            rank = None if tie_rank is None else np.ascontiguousarray(tie_rank, dtype=np.uint32)
            if tie_rank is None and reference_tree == REFTREE_AUTO:
                reference_tree = REFTREE_NEVER   # tie_rank=None means "index order, no reference tree"
        self.tie_rank = rank
        u, v, w = camera_new(eye, look_at, up)
        lt = np.asarray(light_tri, dtype=np.float32).reshape(9)
        d = SceneDesc()
        d.width, d.height = self.width, self.height
        d.eye[:] = _f3(eye).tolist()
        d.u[:], d.v[:], d.w[:] = u.tolist(), v.tolist(), w.tolist()
        d.distance = float(distance)
        d.light_v0[:], d.light_v1[:], d.light_v2[:] = lt[0:3].tolist(), lt[3:6].tolist(), lt[6:9].tolist()
        d.n_tris = len(self.tris)
        d.v0v1v2, d.rgb = _fp(self.tris), _fp(self.rgb)
        d.n_spheres = len(self.spheres)
        if len(self.spheres):
            d.spheres, d.sphere_rgb = _fp(self.spheres), _fp(self.sphere_rgb)
        if self.kinds is not None:
            d.kinds = self.kinds.ctypes.data_as(u8p)
        if rank is not None and len(rank) != self.n_prims:
            raise ValueError("tie_rank must have one entry per primitive")
        d.tie_rank = rank.ctypes.data_as(u32p) if rank is not None else None
        d.nb_ray, d.nb_light_sample = int(nb_ray), int(nb_light_sample)
        d.samples, d.n_samples = _fp(self.samples), len(self.samples)
        d.accel, d.leaf_max, d.reference_tree = int(accel), int(leaf_max), int(reference_tree)
        _check(_lib.rtx_scene_create(C.byref(d), C.byref(self._h)), "rtx_scene_create")

    # -- lifetime
    def close(self):
        if self._h:
            _lib.rtx_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    # -- prepared-scene read-back
    def info(self):
        i = SceneInfo()
        _check(_lib.rtx_scene_info(self._h, C.byref(i)), "rtx_scene_info")
        return i.asdict()

    def light_points(self):
        n = self.info()["n_light_points"]
        out = np.zeros((n, 3), np.float32)
        _check(_lib.rtx_scene_light_points(self._h, _fp(out)), "rtx_scene_light_points")
        return out

    def gamma_thresholds(self):
        out = np.zeros(256, np.float32)
        _check(_lib.rtx_scene_gamma_thresholds(self._h, _fp(out)), "rtx_scene_gamma_thresholds")
        return out

    def normals(self):
        out = np.zeros((self.n_prims, 3), np.float32)      # a sphere's row holds its origin (its normal needs p_hit)
        _check(_lib.rtx_scene_normals(self._h, _fp(out)), "rtx_scene_normals")
        return out

    def nodes(self):
        i = self.info()
        nd = np.zeros((i["n_nodes"], 8), np.uint32)
        order = np.zeros(i["n_tris"], np.uint32)
        _check(_lib.rtx_scene_nodes(self._h, nd.ctypes.data_as(u32p), order.ctypes.data_as(u32p)), "rtx_scene_nodes")
        return nd, order

    def primary_nodes(self):
        """-> (records of the primary rays' stream, whether the scene has one of its own)"""
        nd = np.zeros((self.info()["n_nodes"], 8), np.uint32)
        own = C.c_uint32(0)
        _check(_lib.rtx_scene_primary_nodes(self._h, nd.ctypes.data_as(u32p), C.cast(C.byref(own), u32p)), "rtx_scene_primary_nodes")
        return nd, bool(own.value)

    def ref_nodes(self):
        nd = np.zeros((self.info()["n_ref_nodes"], 8), np.uint32)
        if len(nd):
            _check(_lib.rtx_scene_ref_nodes(self._h, nd.ctypes.data_as(u32p)), "rtx_scene_ref_nodes")
        return nd

    # -- rendering (GPU only)
    def upload(self, device=0):
        _check(_lib.rtx_scene_upload(self._h, device), "rtx_scene_upload")

    def render_rows(self, row0=0, nrows=None, device=0, stats=False, out_ptr=None):
        """out_ptr: address of a caller-owned host buffer of nrows*width*3 bytes (e.g. pinned memory) to render into
        instead of a fresh numpy array; the call then returns None (or the statistics)."""
        if nrows is None:
            nrows = self.height - row0
        out = None if out_ptr else np.zeros((nrows, self.width, 3), np.uint8)
        st = Stats()
        _check(_lib.rtx_render_rows(self._h, device, row0, nrows, C.c_void_p(out_ptr) if out_ptr else out.ctypes.data,
                                    C.byref(st) if stats else None), "rtx_render_rows")
        if out_ptr:
            return st.asdict() if stats else None
        return (out, st.asdict()) if stats else out

    def render_frame(self, devices=(0,), tile_rows=8, stats=False):
        out = np.zeros((self.height, self.width, 3), np.uint8)
        devs = (C.c_int * len(devices))(*devices)
        st = Stats()
        _check(_lib.rtx_render_frame(self._h, devs, len(devices), tile_rows, out.ctypes.data,
                                     C.byref(st) if stats else None), "rtx_render_frame")
        return (out, st.asdict()) if stats else out

    def wave_profile(self, row0=0, nrows=None, device=0):
        """Diagnostics: [tiles_y, tiles_x, 8] uint64 {node fetches, triangle fetches, start, end, primary phase,
        shadow phase (slowest wavefront), accumulation phase, -}; times in 100 MHz ticks.  Ablation build only
        (RTX_PY_ABLATION=1): librtx.so answers ERR_UNSUPPORTED."""
        if nrows is None:
            nrows = self.height - row0
        tx, ty = C.c_uint32(), C.c_uint32()
        _check(_lib.rtx_debug_wave_profile(self._h, device, row0, nrows, None, 0, C.byref(tx), C.byref(ty)),
               "rtx_debug_wave_profile")
        out = np.zeros((ty.value, tx.value, 8), np.uint64)
        _check(_lib.rtx_debug_wave_profile(self._h, device, row0, nrows, out.ctypes.data_as(u64p), tx.value * ty.value,
                                           C.byref(tx), C.byref(ty)), "rtx_debug_wave_profile")
        return out

    def launch_timings(self, device=0, max_launches=64):
        """Device milliseconds of the most recent launches, oldest first: (scheduling pass, shading pass) arrays."""
        a = np.zeros(max_launches, np.float32)
        b = np.zeros(max_launches, np.float32)
        n = _lib.rtx_launch_timings(self._h, device, max_launches, _fp(a), _fp(b))
        if n < 0:
            raise RtxError(n, "rtx_launch_timings")
        return a[:n], b[:n]

    def tile_descs(self, device=0):
        """Diagnostics: uint32 [tiles, 4] {cost class, primary hits, flags, reserved} of the most recent launch."""
        n = _lib.rtx_debug_tile_descs(self._h, device, None, 0)
        if n < 0:
            raise RtxError(n, "rtx_debug_tile_descs")
        out = np.zeros((n, 4), np.uint32)
        m = _lib.rtx_debug_tile_descs(self._h, device, out.ctypes.data_as(u32p), n)
        if m < 0:
            raise RtxError(m, "rtx_debug_tile_descs")
        return out[:m]

    def tiles_rows(self, first_tile, tile_stride, tile_rows):
        return _lib.rtx_tiles_rows(self._h, first_tile, tile_stride, tile_rows)

    def tiles_bytes(self, first_tile, tile_stride, tile_rows):
        return _lib.rtx_tiles_bytes(self._h, first_tile, tile_stride, tile_rows)

    def render_tiles_device(self, device, first_tile, tile_stride, tile_rows, d_out_ptr, d_out_bytes,
                            stream=None, d_counters_ptr=None):
        """Asynchronous launch into a device buffer the caller owns (e.g. a torch uint8 tensor)."""
        _check(_lib.rtx_render_tiles_device(self._h, device, first_tile, tile_stride, tile_rows,
                                            C.c_void_p(d_out_ptr), d_out_bytes,
                                            C.c_void_p(stream) if stream else None,
                                            C.c_void_p(d_counters_ptr) if d_counters_ptr else None),
               "rtx_render_tiles_device")


def default_scene(obj_paths, width=DEFAULT_WIDTH, height=DEFAULT_HEIGHT, samples=None, **kw):
    """The scene main() renders (src/main.rs:319-362) for the given OBJ files."""
    tris, rgb = default_primitives(obj_paths)
    if samples is None:
        samples = gen_samples()
    return Scene(width, height, tris, rgb, samples, **kw)


def tile_owner(tile, world_size):
    """Row tile -> rank (interleaved): tile t belongs to rank t % world_size."""
    return tile % world_size


def scatter_tiles(frame, packed, first_tile, tile_stride, tile_rows):
    """Place the packed rows of rtx_render_tiles_device back into a [H, W, 3] frame (rtxh_scatter_tiles: the loop
    rtx_render_frame itself gathers with)."""
    h, w = frame.shape[0], frame.shape[1]
    assert frame.flags["C_CONTIGUOUS"] and frame.dtype == np.uint8
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    assert packed.size == tiles_rows_of(h, first_tile, tile_stride, tile_rows) * w * 3, "packed share has the wrong size"
    _check(_lib.rtxh_scatter_tiles(frame.ctypes.data, h, w, packed.ctypes.data, first_tile, tile_stride, tile_rows),
           "rtxh_scatter_tiles")
    return frame


def tiles_rows_of(height, first_tile, tile_stride, tile_rows):
    """Rows of the share (first_tile, tile_stride, tile_rows) of a frame of `height` rows."""
    rows, t = 0, first_tile
    while t * tile_rows < height:
        rows += min(tile_rows, height - t * tile_rows)
        t += tile_stride
    return rows
