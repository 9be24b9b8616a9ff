"""ctypes loader for the CPU oracle (liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BN_BATCH = 1
HIDDEN_CARRY = 2


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def _bind_dn(so):
    L = C.CDLL(so)
    L.orc_dn_create.restype = C.c_void_p
    L.orc_dn_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    L.orc_dn_destroy.argtypes = [C.c_void_p]
    L.orc_dn_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
    L.orc_dn_get_hidden.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.orc_dn_set_hidden.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.orc_dn_reset_hidden.argtypes = [C.c_void_p]
    L.orc_dn_conv_checksums.argtypes = [C.c_void_p, C.c_void_p]
    return L


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = _bind_dn(so)
    return _LIB


_LIB64 = None


def lib64():
    """the denoiser restatement compiled with -DORC_DN_FP64: double activations, products and sums (liboracle64.so)"""
    global _LIB64
    if _LIB64 is None:
        so = os.path.join(_HERE, "liboracle64.so")
        if not os.path.exists(so):
            build(force=True)
        _LIB64 = _bind_dn(so)
    return _LIB64


class DenoiseOracle:
    """CPU restatement of AutoEncoder.forward (reference: recurrent_autoencoder_model.py:120-142)."""

    def __init__(self, blob: bytes, H: int, W: int, fp64: bool = False):
        """fp64: the all-double build (the truth of the precision studies); the interface stays float32"""
        from ai_path_tracer_denoiser_amd import arch
        self.H, self.W = H, W
        self._arch = arch
        self._blob = blob
        self._L = lib64() if fp64 else lib()
        self._h = self._L.orc_dn_create(blob, len(blob), H, W)
        if not self._h:
            raise ValueError("orc_dn_create failed (bad blob, or H/W not multiples of 32)")

    def close(self):
        if self._h:
            self._L.orc_dn_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def forward(self, x10: np.ndarray, bn_batch: bool, carry: bool) -> np.ndarray:
        x = np.ascontiguousarray(x10, dtype=np.float32)
        assert x.shape == (10, self.H, self.W)
        out = np.empty((3, self.H, self.W), np.float32)
        flags = (BN_BATCH if bn_batch else 0) | (HIDDEN_CARRY if carry else 0)
        rc = self._L.orc_dn_forward(self._h, x.ctypes.data, out.ctypes.data, flags)
        if rc:
            raise RuntimeError(f"orc_dn_forward rc={rc}")
        return out

    def hidden(self, level: int) -> np.ndarray:
        shp = self._arch.hidden_shapes(self.H, self.W)[level]
        h = np.empty(shp, np.float32)
        self._L.orc_dn_get_hidden(self._h, level, h.ctypes.data)
        return h

    def set_hidden(self, level: int, h: np.ndarray):
        h = np.ascontiguousarray(h, dtype=np.float32)
        assert h.shape == self._arch.hidden_shapes(self.H, self.W)[level]
        self._L.orc_dn_set_hidden(self._h, level, h.ctypes.data)

    def reset_hidden(self):
        self._L.orc_dn_reset_hidden(self._h)

    def conv_checksums(self) -> np.ndarray:
        s = np.empty(28, np.float64)
        self._L.orc_dn_conv_checksums(self._h, s.ctypes.data)
        return s


# --------------------------------------------------------------------------- trace oracle
class Geom(C.Structure):          # sceneStructs.h:20 (248 B)
    _fields_ = [("type", C.c_int), ("materialid", C.c_int),
                ("translation", C.c_float * 3), ("rotation", C.c_float * 3), ("scale", C.c_float * 3),
                ("transform", C.c_float * 16), ("inverseTransform", C.c_float * 16),
                ("invTranspose", C.c_float * 16), ("vel", C.c_float * 3)]


class Face(C.Structure):          # sceneStructs.h:40 (76 B)
    _fields_ = [("v", (C.c_float * 3) * 3), ("n", (C.c_float * 3) * 3), ("materialid", C.c_int)]


class Material(C.Structure):      # sceneStructs.h:46 (44 B)
    _fields_ = [("color", C.c_float * 3), ("spec_exponent", C.c_float), ("spec_color", C.c_float * 3),
                ("hasReflective", C.c_float), ("hasRefractive", C.c_float),
                ("indexOfRefraction", C.c_float), ("emittance", C.c_float)]


class Camera(C.Structure):        # sceneStructs.h:58 (84 B)
    _fields_ = [("res", C.c_int * 2), ("position", C.c_float * 3), ("lookAt", C.c_float * 3),
                ("view", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("fov", C.c_float * 2), ("pixelLength", C.c_float * 2)]


class AABB(C.Structure):          # sceneStructs.h:84 (24 B)
    _fields_ = [("lb", C.c_float * 3), ("ub", C.c_float * 3)]


assert (C.sizeof(Geom), C.sizeof(Face), C.sizeof(Material), C.sizeof(Camera), C.sizeof(AABB)) == (248, 76, 44, 84, 24)

SPHERE, CUBE = 0, 1
TRACE_AA, TRACE_COMPACT, TRACE_SORT_MATERIAL, TRACE_CACHE_FIRST_BOUNCE = 1, 2, 32, 64     # = include/aiptd.h AIPT_TRACE_*
TRACE_ORACLE_BVH = 256       # oracle only: the face loop through a CPU BVH (oracle/trace_bvh.c), same result
TRACE_NO_CULL, TRACE_DIELECTRIC, TRACE_MESH_NORMAL_VIEW = 512, 1024, 2048   # RAY_CULLING false, DIELECTRIC true, MESH_NORMAL_VIEW true


def _trace_lib():
    L = lib()
    if not getattr(L, "_trace_ready", False):
        L.orc_utilhash.restype = C.c_uint32
        L.orc_utilhash.argtypes = [C.c_uint32]
        L.orc_seed.restype = C.c_uint32
        L.orc_seed.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_lcg_next.restype = C.c_uint32
        L.orc_lcg_next.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_u01.restype = C.c_float
        L.orc_u01.argtypes = [C.POINTER(C.c_uint32), C.c_float, C.c_float]
        L.orc_det_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_hemisphere.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
        for f in (L.orc_box_test, L.orc_sphere_test):
            f.restype = C.c_float
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_triangle_test.restype = C.c_float
        L.orc_triangle_test.argtypes = [C.c_void_p] * 5
        L.orc_ray_aabb.restype = C.c_int
        L.orc_ray_aabb.argtypes = [C.c_void_p] * 3
        L.orc_scatter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_scatter_dielectric.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_fresnel_dielectric.restype = C.c_float
        L.orc_fresnel_dielectric.argtypes = [C.c_float, C.c_float, C.c_float]
        L.orc_intersect_scene.restype = C.c_float
        L.orc_intersect_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                          C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        L.orc_pathtrace.restype = C.c_int
        L.orc_pathtrace.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_pathtrace_accum.restype = C.c_int
        L.orc_pathtrace_accum.argtypes = L.orc_pathtrace.argtypes + [C.c_void_p]
        L.orc_pathtrace_ex.restype = C.c_int
        L.orc_pathtrace_ex.argtypes = L.orc_pathtrace.argtypes[:10] + [C.c_uint] + L.orc_pathtrace.argtypes[10:] + [C.c_void_p] * 2
        L.orc_move_geoms.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.orc_build_geom.argtypes = [C.c_void_p]
        L.orc_camera_setup.argtypes = [C.c_void_p, C.c_float]
        L.orc_camera_orbit_params.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 3
        L.orc_camera_orbit.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L._trace_ready = True
    return L


class OracleScene:
    """Scene arrays for the trace oracle.  ``parse`` is a small test-side reader of the reference's
    scene grammar (scene.cpp:21-40, 44-100, 102-159, 161-196) -- independent of the product's C++ parser."""

    def __init__(self):
        self.geoms, self.materials, self.faces = [], [], []
        self.camera = Camera()
        self.mesh_box = AABB()
        self.fovy = 45.0
        self.depth = 8
        self.iterations = 1

    @staticmethod
    def parse(path, res=None, depth=None):
        L = _trace_lib()
        s = OracleScene()
        with open(path) as f:
            lines = [ln.strip() for ln in f.read().replace("\r\n", "\n").split("\n")]
        i = 0

        def nxt():
            nonlocal i
            ln = lines[i] if i < len(lines) else ""
            i += 1
            return ln

        while i < len(lines):
            ln = nxt()
            tok = ln.split()
            if not tok:
                continue
            if tok[0] == "MATERIAL":
                m = Material()
                for _ in range(7):                      # exactly 7 lines (scene.cpp:171)
                    t = nxt().split()
                    if t[0] == "RGB":
                        m.color[:] = [float(v) for v in t[1:4]]
                    elif t[0] == "SPECEX":
                        m.spec_exponent = float(t[1])
                    elif t[0] == "SPECRGB":
                        m.spec_color[:] = [float(v) for v in t[1:4]]
                    elif t[0] == "REFL":
                        m.hasReflective = float(t[1])
                    elif t[0] == "REFR":
                        m.hasRefractive = float(t[1])
                    elif t[0] == "REFRIOR":
                        m.indexOfRefraction = float(t[1])
                    elif t[0] == "EMITTANCE":
                        m.emittance = float(t[1])
                s.materials.append(m)
            elif tok[0] == "OBJECT":
                g = Geom()
                kind = nxt()
                g.type = SPHERE if kind == "sphere" else CUBE
                g.materialid = int(nxt().split()[1])
                ln2 = nxt()
                while ln2:
                    t = ln2.split()
                    vals = [float(v) for v in t[1:4]]
                    if t[0] == "TRANS":
                        g.translation[:] = vals
                    elif t[0] == "ROTAT":
                        g.rotation[:] = vals
                    elif t[0] == "SCALE":
                        g.scale[:] = vals
                    elif t[0] == "VEL":
                        g.vel[:] = vals
                    ln2 = nxt()
                L.orc_build_geom(C.byref(g))
                s.geoms.append(g)
            elif tok[0] == "CAMERA":
                cam = s.camera
                for _ in range(5):                      # exactly 5 lines (scene.cpp:109)
                    t = nxt().split()
                    if t[0] == "RES":
                        cam.res[:] = [int(t[1]), int(t[2])]
                    elif t[0] == "FOVY":
                        s.fovy = float(t[1])
                    elif t[0] == "ITERATIONS":
                        s.iterations = int(t[1])
                    elif t[0] == "DEPTH":
                        s.depth = int(t[1])
                ln2 = nxt()
                while ln2:
                    t = ln2.split()
                    vals = [float(v) for v in t[1:4]]
                    if t[0] == "EYE":
                        cam.position[:] = vals
                    elif t[0] == "LOOKAT":
                        cam.lookAt[:] = vals
                    elif t[0] == "UP":
                        cam.up[:] = vals
                    ln2 = nxt()
        if res is not None:
            s.camera.res[:] = list(res)
        if depth is not None:
            s.depth = depth
        L.orc_camera_setup(C.byref(s.camera), C.c_float(s.fovy))
        z, p, t = C.c_float(), C.c_float(), C.c_float()
        L.orc_camera_orbit_params(C.byref(s.camera), C.byref(z), C.byref(p), C.byref(t))
        s.zoom, s.phi, s.theta = z.value, p.value, t.value
        s.set_orbit(s.zoom, s.phi, s.theta)
        return s

    def set_orbit(self, zoom, phi, theta):
        """runCuda() camera rebuild (main.cpp:122-140)."""
        _trace_lib().orc_camera_orbit(C.byref(self.camera), C.c_float(zoom), C.c_float(phi), C.c_float(theta))

    def set_mesh(self, faces_np, lb, ub):
        """Attach a mesh given as a numpy structured array with the 76-byte Face layout (synth.FACE_DTYPE)."""
        assert faces_np.dtype.itemsize == 76
        self.faces_np = np.ascontiguousarray(faces_np)
        self.mesh_box.lb[:] = [float(v) for v in lb]
        self.mesh_box.ub[:] = [float(v) for v in ub]

    @property
    def nfaces(self):
        return len(self.faces_np) if getattr(self, "faces_np", None) is not None else len(self.faces)

    def arrays(self):
        ga = (Geom * max(1, len(self.geoms)))(*self.geoms)
        ma = (Material * max(1, len(self.materials)))(*self.materials)
        if getattr(self, "faces_np", None) is not None:
            fa = self.faces_np.ctypes.data_as(C.c_void_p)
        else:
            fa = (Face * max(1, len(self.faces)))(*self.faces)
        return ga, ma, fa

    def move_geoms(self, dt=0.10):
        """moveGeom (pathtrace.cu:318-331): translate the primitives that have a velocity, rebuild their matrices."""
        ga = (Geom * max(1, len(self.geoms)))(*self.geoms)
        _trace_lib().orc_move_geoms(ga, len(self.geoms), C.c_float(dt))
        self.geoms = [Geom.from_buffer_copy(bytes(ga[i])) for i in range(len(self.geoms))]

    def pathtrace(self, iter=1, depth=None, pad_rows_to=None, want_mat0=True, accum=None, gbuf=None,
                  flags=TRACE_AA | TRACE_COMPACT, cache=None):
        """One 1-spp frame (pathtrace.cu:422-528).  Returns (gbuf[10,Hp,W], n_live[depth+1], mat0[H*W]).
        flags: TRACE_* below (the reference's #defines, pathtrace.cu:20-26); cache: uint8[W*H*36] kept by the caller
        across iterations for TRACE_CACHE_FIRST_BOUNCE."""
        L = _trace_lib()
        W, H = self.camera.res[0], self.camera.res[1]
        Hp = pad_rows_to or H
        depth = depth or self.depth
        if gbuf is None:
            gbuf = np.zeros((10, Hp, W), np.float32)
        n_live = np.full(depth + 1, -1, np.int32)
        mat0 = np.full(W * H, -2, np.int32)
        ga, ma, fa = self.arrays()
        nb = L.orc_pathtrace_ex(C.byref(self.camera), ga, len(self.geoms), ma, len(self.materials), fa,
                                self.nfaces, C.byref(self.mesh_box), iter, depth, flags, gbuf.ctypes.data, Hp,
                                n_live.ctypes.data, mat0.ctypes.data if want_mat0 else None,
                                accum.ctypes.data if accum is not None else None,
                                cache.ctypes.data if cache is not None else None)
        return gbuf, n_live[:nb + 1], mat0
