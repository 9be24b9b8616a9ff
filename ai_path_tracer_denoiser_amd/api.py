"""ctypes binding of libaiptd.so (include/aiptd.h) and a thin host-side mirror of the
reference's interface for the hot path.

Reference interface mirrored (Inference/src):
    pathtraceInit(Scene*) / pathtrace(pbo, frame, iter) / pathtraceFree()   pathtrace.h:6-8
    torch::jit::load + module.forward                                        main.cpp:104-111
    runCuda() per-frame body                                                 main.cpp:143-163

torch is used only as plumbing (device buffers, stream handles); every computation happens in
the HIP library.  There is no fallback: if libaiptd.so is missing or no GPU is visible, the
calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AIPT_LIB") or os.path.join(_HERE, "libaiptd.so")   # AIPT_LIB: kernel-variant builds (tools/)

# flags (include/aiptd.h)
TRACE_AA, TRACE_COMPACT, TRACE_RECORD_MAT0, TRACE_BRUTE_FORCE, TRACE_NO_BROAD_PHASE = 1, 2, 4, 8, 16
TRACE_SORT_MATERIAL, TRACE_CACHE_FIRST_BOUNCE, TRACE_MOTION_BLUR = 32, 64, 128
TRACE_NO_CULL, TRACE_DIELECTRIC, TRACE_MESH_NORMAL_VIEW = 512, 1024, 2048
SCENE_RECOMPUTE_NORMALS = 1
TRACE_DEFAULT = TRACE_AA | TRACE_COMPACT
DN_BN_BATCH, DN_BN_RUNNING, DN_HIDDEN_CARRY, DN_HIDDEN_RESET = 1, 0, 2, 0
DN_IMPL_MFMA, DN_IMPL_VALU, DN_IMPL_MFMA_F16X3, DN_IMPL_MFMA_F16W = 0, 1, 2, 3
DN_OPT_R_MINPIX, DN_OPT_F16_MINPIX, DN_OPT_SMALL_MINPIX, DN_OPT_FUSED_POOL, DN_OPT_KY_SPLIT = 1, 2, 3, 4, 5
GEOM_SPHERE, GEOM_CUBE = 0, 1


class Geom(C.Structure):
    _fields_ = [("type", C.c_int), ("materialid", C.c_int),
                ("translation", C.c_float * 3), ("rotation", C.c_float * 3), ("scale", C.c_float * 3),
                ("transform", C.c_float * 16), ("inverseTransform", C.c_float * 16),
                ("invTranspose", C.c_float * 16), ("vel", C.c_float * 3)]


class Face(C.Structure):
    _fields_ = [("v", (C.c_float * 3) * 3), ("n", (C.c_float * 3) * 3), ("materialid", C.c_int)]


class Material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("specular_exponent", C.c_float), ("specular_color", C.c_float * 3),
                ("hasReflective", C.c_float), ("hasRefractive", C.c_float),
                ("indexOfRefraction", C.c_float), ("emittance", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("resolution", C.c_int * 2), ("position", C.c_float * 3), ("lookAt", C.c_float * 3),
                ("view", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("fov", C.c_float * 2), ("pixelLength", C.c_float * 2)]


class AABB(C.Structure):
    _fields_ = [("lb", C.c_float * 3), ("ub", C.c_float * 3)]


assert (C.sizeof(Geom), C.sizeof(Face), C.sizeof(Material), C.sizeof(Camera), C.sizeof(AABB)) == (248, 76, 44, 84, 24)

# every symbol include/aiptd.h declares: (name, restype, argtypes)
_P = C.c_void_p
ABI = [
    ("aipt_create", C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    ("aipt_destroy", None, [_P]),
    ("aipt_last_error", C.c_char_p, [_P]),
    ("aipt_abi_version", C.c_int, []),
    ("aipt_sync", C.c_int, [_P]),
    ("aipt_malloc", C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    ("aipt_free", C.c_int, [_P, _P]),
    ("aipt_upload", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("aipt_download", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("aipt_memset", C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    ("aipt_timer_start", C.c_int, [_P]),
    ("aipt_timer_stop", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("aipt_scene_upload", C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P]),
    ("aipt_scene_free", C.c_int, [_P]),
    ("aipt_scene_pack", C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.POINTER(_P), C.POINTER(C.c_size_t), C.c_char_p,
                                  C.c_size_t]),
    ("aipt_blob_free", None, [_P]),
    ("aipt_scene_upload_packed", C.c_int, [_P, _P, C.c_size_t]),
    ("aipt_trace_configure_batch", C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    ("aipt_trace_batch", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_uint32, _P, C.c_int, C.c_int, C.c_size_t]),
    ("aipt_trace_live_counts_frame", C.c_int, [_P, C.c_int, _P, C.c_int]),
    ("aipt_frames_configure", C.c_int, [_P, C.c_int]),
    ("aipt_frames", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    ("aipt_frames_prefetch", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_uint32]),
    ("aipt_frames_gbuffer", C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("aipt_trace_profile_begin", C.c_int, [_P, C.c_int, C.c_int]),
    ("aipt_trace_profile_end", C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    ("aipt_trace_profile_calls", C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    ("aipt_trace_kernel_name", C.c_int, [_P, C.c_int, C.c_char_p, C.c_size_t]),
    ("aipt_trace_configure", C.c_int, [_P, C.c_int, C.c_int]),
    ("aipt_trace", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_uint32, _P, C.c_int, C.c_int]),
    ("aipt_debug_trace_stats", C.c_int, [_P, _P, C.c_int]),
    ("aipt_trace_live_counts", C.c_int, [_P, _P, C.c_int]),
    ("aipt_trace_first_hit_materials", C.c_int, [_P, _P, C.c_int]),
    ("aipt_denoise_load_weights", C.c_int, [_P, _P, C.c_size_t]),
    ("aipt_denoise_configure", C.c_int, [_P, C.c_int, C.c_int]),
    ("aipt_denoise_set_impl", C.c_int, [_P, C.c_int]),
    ("aipt_denoise_set_option", C.c_int, [_P, C.c_int, C.c_longlong]),
    ("aipt_denoise", C.c_int, [_P, _P, _P, C.c_uint32]),
    ("aipt_denoise_reset_hidden", C.c_int, [_P]),
    ("aipt_denoise_get_hidden", C.c_int, [_P, C.c_int, _P]),
    ("aipt_denoise_set_hidden", C.c_int, [_P, C.c_int, _P]),
    ("aipt_denoise_profile_begin", C.c_int, [_P, C.c_uint32, C.c_int]),
    ("aipt_denoise_profile_stride", C.c_int, [_P, C.c_int]),
    ("aipt_denoise_profile_end", C.c_int, [_P, _P, C.POINTER(C.c_int)]),
    ("aipt_denoise_layer_info", C.c_int, [_P, C.c_int, C.c_char_p, C.c_size_t] + [C.POINTER(C.c_int)] * 4
     + [C.POINTER(C.c_double)]),
    ("aipt_frame_configure", C.c_int, [_P, C.c_int, C.c_int]),
    ("aipt_frame", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_uint32, C.c_uint32, _P]),
    ("aipt_gbuffer", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("aipt_frame_set_timing", C.c_int, [_P, C.c_int]),
    ("aipt_frame_last_times", C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("aipt_frame_prefetch", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_uint32]),
    ("aipt_comm_create", C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.POINTER(_P)]),
    ("aipt_comm_is_rccl", C.c_int, [_P]),
    ("aipt_comm_broadcast", C.c_int, [_P, C.POINTER(_P), C.c_size_t, C.c_int]),
    ("aipt_comm_destroy", None, [_P]),
    ("aipt_device_count", C.c_int, []),
    ("aipt_scene_load", C.c_int, [C.c_char_p, C.POINTER(_P), C.c_char_p, C.c_size_t]),
    ("aipt_scene_load_ex", C.c_int, [C.c_char_p, C.c_uint, C.POINTER(_P), C.c_char_p, C.c_size_t]),
    ("aipt_scene_release", None, [_P]),
    ("aipt_scene_set_resolution", C.c_int, [_P, C.c_int, C.c_int]),
    ("aipt_scene_info", C.c_int, [_P] + [C.POINTER(C.c_int)] * 5),
    ("aipt_scene_geoms", C.POINTER(Geom), [_P]),
    ("aipt_scene_materials", C.POINTER(Material), [_P]),
    ("aipt_scene_faces", C.POINTER(Face), [_P]),
    ("aipt_scene_mesh_box", C.POINTER(AABB), [_P]),
    ("aipt_scene_camera", C.c_int, [_P, C.POINTER(Camera)]),
    ("aipt_scene_orbit_params", C.c_int, [_P] + [C.POINTER(C.c_float)] * 3),
    ("aipt_scene_upload_host", C.c_int, [_P, _P]),
    ("aipt_geom_build", None, [C.POINTER(Geom)]),
    ("aipt_camera_orbit_params", None, [C.POINTER(Camera)] + [C.POINTER(C.c_float)] * 3),
    ("aipt_camera_orbit", None, [C.POINTER(Camera), C.c_float, C.c_float, C.c_float]),
]

_LIB = None


class AiptError(RuntimeError):
    pass


def lib():
    """Load libaiptd.so; raises if it has not been built (no fallback path exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise AiptError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950); this package has no CPU or PyTorch fallback")
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  Import it first so
        # that libaiptd.so's DT_NEEDED libamdhip64.so.7 resolves to that already-loaded copy: two HIP runtimes in one
        # process leave the second one without a GPU ("No HIP GPUs are available").
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, res, args in ABI:
            fn = getattr(L, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def scene_pack(geoms, materials, faces=(), mesh_box=None) -> bytes:
    """aipt_scene_pack: validate, build the mesh BVH once (host only) and return the relocatable scene blob that
    Context.pathtrace_init_packed uploads -- the unit rank 0 broadcasts to the other GPUs."""
    L = lib()
    ga = (Geom * max(1, len(geoms)))(*geoms)
    ma = (Material * max(1, len(materials)))(*materials)
    if isinstance(faces, np.ndarray):
        assert faces.dtype.itemsize == 76
        keep = np.ascontiguousarray(faces)
        fa = keep.ctypes.data_as(_P)
    else:
        fa = (Face * max(1, len(faces)))(*faces)
    box = mesh_box if mesh_box is not None else AABB()
    out, n = _P(), C.c_size_t()
    err = C.create_string_buffer(256)
    rc = L.aipt_scene_pack(ga, len(geoms), ma, len(materials), fa if len(faces) else None, len(faces),
                           C.byref(box) if len(faces) else None, C.byref(out), C.byref(n), err, 256)
    if rc:
        raise AiptError(f"aipt_scene_pack failed ({rc}): {err.value.decode()}")
    try:
        return C.string_at(out.value, n.value)
    finally:
        L.aipt_blob_free(out)


class Scene:
    """Host-side scene (reference class Scene, scene.h:13-44): parsed by the C++ front end of libaiptd.so."""

    def __init__(self, path: str, res=None, depth=None, flags=0):
        L = lib()
        h = _P()
        err = C.create_string_buffer(512)
        rc = L.aipt_scene_load_ex(path.encode(), int(flags), C.byref(h), err, 512)   # flags: SCENE_RECOMPUTE_NORMALS
        if rc:
            raise AiptError(f"aipt_scene_load({path}) failed ({rc}): {err.value.decode()}")
        self._h = h
        if res is not None:
            rc = L.aipt_scene_set_resolution(h, int(res[0]), int(res[1]))
            if rc:
                raise AiptError(f"aipt_scene_set_resolution{tuple(res)} failed ({rc})")
        n = [C.c_int() for _ in range(5)]
        L.aipt_scene_info(h, *[C.byref(v) for v in n])
        self.ngeoms, self.nmaterials, self.nfaces, self.iterations, self.depth = [v.value for v in n]
        if depth is not None:
            self.depth = depth
        self.camera = Camera()
        L.aipt_scene_camera(h, C.byref(self.camera))
        z, p, t = C.c_float(), C.c_float(), C.c_float()
        L.aipt_scene_orbit_params(h, C.byref(z), C.byref(p), C.byref(t))
        self.zoom, self.phi, self.theta = z.value, p.value, t.value

    def close(self):
        if getattr(self, "_h", None):
            lib().aipt_scene_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def geoms(self):
        p = lib().aipt_scene_geoms(self._h)
        return [p[i] for i in range(self.ngeoms)]

    @property
    def materials(self):
        p = lib().aipt_scene_materials(self._h)
        return [p[i] for i in range(self.nmaterials)]

    @property
    def faces(self):
        p = lib().aipt_scene_faces(self._h)
        return [p[i] for i in range(self.nfaces)]

    @property
    def mesh_box(self):
        return lib().aipt_scene_mesh_box(self._h).contents

    def orbit(self, zoom=None, phi=None, theta=None) -> Camera:
        """runCuda() camera rebuild (main.cpp:122-140); returns a Camera for this frame."""
        cam = Camera.from_buffer_copy(bytes(self.camera))
        lib().aipt_camera_orbit(C.byref(cam), self.zoom if zoom is None else zoom,
                                self.phi if phi is None else phi, self.theta if theta is None else theta)
        return cam


class Context:
    """One context per GPU (aipt_create).  stream: a raw hipStream_t handle (int) or None."""

    def __init__(self, device: int = 0, stream=None):
        L = lib()
        h = _P()
        rc = L.aipt_create(device, _P(stream) if stream else None, C.byref(h))
        if rc:
            raise AiptError(f"aipt_create failed ({rc}): {L.aipt_last_error(None).decode()}")
        self._h = h
        self.device = device
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            lib().aipt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise AiptError(f"libaiptd error {rc}: {lib().aipt_last_error(self._h).decode()}")

    def sync(self):
        self._ck(lib().aipt_sync(self._h))

    def timer_start(self):
        self._ck(lib().aipt_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._ck(lib().aipt_timer_stop(self._h, C.byref(ms)))
        return ms.value

    # ------------------------------------------------------------------ trace (pathtrace.h:6-8)
    def pathtrace_init(self, geoms, materials, faces=(), mesh_box=None, width=None, height=None):
        """pathtraceInit(Scene*): upload the scene; allocate path state for width x height."""
        ga = (Geom * max(1, len(geoms)))(*geoms)
        ma = (Material * max(1, len(materials)))(*materials)
        if isinstance(faces, np.ndarray):                      # structured array with the 76-byte Face layout
            assert faces.dtype.itemsize == 76
            keep = np.ascontiguousarray(faces)
            fa = keep.ctypes.data_as(_P)
        else:
            fa = (Face * max(1, len(faces)))(*faces)
        box = mesh_box if mesh_box is not None else AABB()
        self._ck(lib().aipt_scene_upload(self._h, ga, len(geoms), ma, len(materials),
                                         fa if len(faces) else None, len(faces),
                                         C.byref(box) if len(faces) else None))
        if width is not None:
            self._ck(lib().aipt_trace_configure(self._h, width, height))

    def pathtrace_init_packed(self, blob: bytes, width=None, height=None):
        """pathtraceInit from a packed scene (scene_pack): copies only, the BVH inside the blob is not rebuilt."""
        self._ck(lib().aipt_scene_upload_packed(self._h, blob, len(blob)))
        if width is not None:
            self._ck(lib().aipt_trace_configure(self._h, width, height))

    def trace_profile_begin(self, max_calls: int, every: int = 1):
        self._ck(lib().aipt_trace_profile_begin(self._h, max_calls, every))

    def trace_profile_end(self, nbounces: int):
        """-> (summed ms of every bounce launch [nbounces], number of recorded traces)"""
        ms = np.zeros(nbounces, np.float64)
        n = C.c_int()
        self._ck(lib().aipt_trace_profile_end(self._h, ms.ctypes.data, nbounces, C.byref(n)))
        return ms, n.value

    def trace_profile_calls(self, nbounces: int, max_calls: int = 4096):
        """-> (frames held by every recorded trace call [calls], ms of its bounce launches [calls, nbounces]); before _end"""
        fr = np.zeros(max_calls, np.int32)
        ms = np.zeros((max_calls, nbounces), np.float64)
        n = C.c_int()
        self._ck(lib().aipt_trace_profile_calls(self._h, fr.ctypes.data, ms.ctypes.data, nbounces, max_calls, C.byref(n)))
        return fr[:n.value].copy(), ms[:n.value].copy()

    def trace_kernel_name(self, bounce: int) -> str:
        name = C.create_string_buffer(64)
        self._ck(lib().aipt_trace_kernel_name(self._h, bounce, name, 64))
        return name.value.decode()

    def pathtrace_init_scene(self, scene: "Scene", width=None, height=None):
        """pathtraceInit(Scene*) from a parsed Scene."""
        self._ck(lib().aipt_scene_upload_host(self._h, scene._h))
        if width is not None:
            self._ck(lib().aipt_trace_configure(self._h, width, height))

    def pathtrace_free(self):
        self._ck(lib().aipt_scene_free(self._h))

    def pathtrace(self, cam: Camera, iter: int, depth: int, gbuf, flags: int = TRACE_DEFAULT):
        """pathtrace(pbo, frame, iter): one 1-spp iteration into gbuf, a float32 device tensor [10, rows, stride]."""
        assert gbuf.is_cuda and gbuf.dtype.is_floating_point and gbuf.dim() == 3 and gbuf.shape[0] == 10
        assert gbuf.is_contiguous()
        self._ck(lib().aipt_trace(self._h, C.byref(cam), iter, depth, flags, _P(gbuf.data_ptr()),
                                  gbuf.shape[1], gbuf.shape[2]))

    def live_counts(self, depth: int) -> np.ndarray:
        out = np.zeros(depth + 1, np.int32)
        self._ck(lib().aipt_trace_live_counts(self._h, out.ctypes.data, depth + 1))
        return out

    def first_hit_materials(self, n: int) -> np.ndarray:
        out = np.zeros(n, np.int32)
        self._ck(lib().aipt_trace_first_hit_materials(self._h, out.ctypes.data, n))
        return out

    # ------------------------------------------------------------------ denoiser (main.cpp:101-118)
    def load_weights(self, blob: bytes):
        """torch::jit::load(MODEL_PATH) counterpart: flat blob (arch.py)."""
        self._ck(lib().aipt_denoise_load_weights(self._h, blob, len(blob)))

    def denoise_configure(self, H: int, W: int):
        self._ck(lib().aipt_denoise_configure(self._h, H, W))

    def denoise_set_impl(self, impl: int):
        self._ck(lib().aipt_denoise_set_impl(self._h, impl))

    def denoise_set_option(self, option, value):
        """aipt_denoise_set_option: DN_OPT_R_MINPIX / DN_OPT_F16_MINPIX / DN_OPT_SMALL_MINPIX / DN_OPT_FUSED_POOL / DN_OPT_KY_SPLIT"""
        self._ck(lib().aipt_denoise_set_option(self._h, option, value))

    def denoise(self, x10, out3, bn_batch: bool = True, carry: bool = False):
        """module.forward: x10 [10,H,W] -> out3 [3,H,W], float32 contiguous device tensors."""
        assert x10.is_cuda and out3.is_cuda and x10.is_contiguous() and out3.is_contiguous()
        flags = (DN_BN_BATCH if bn_batch else 0) | (DN_HIDDEN_CARRY if carry else 0)
        self._ck(lib().aipt_denoise(self._h, _P(x10.data_ptr()), _P(out3.data_ptr()), flags))

    def reset_hidden(self):
        self._ck(lib().aipt_denoise_reset_hidden(self._h))

    def get_hidden(self, level: int, dst):
        self._ck(lib().aipt_denoise_get_hidden(self._h, level, _P(dst.data_ptr())))

    def set_hidden(self, level: int, src):
        self._ck(lib().aipt_denoise_set_hidden(self._h, level, _P(src.data_ptr())))

    def profile_begin(self, layer_mask: int, max_calls: int):
        self._ck(lib().aipt_denoise_profile_begin(self._h, layer_mask, max_calls))

    def profile_end(self):
        """-> (summed ms per layer [28], number of recorded forwards)"""
        ms = np.zeros(28, np.float64)
        n = C.c_int()
        self._ck(lib().aipt_denoise_profile_end(self._h, ms.ctypes.data, C.byref(n)))
        return ms, n.value

    def profile_stride(self, every: int):
        self._ck(lib().aipt_denoise_profile_stride(self._h, every))

    def layer_info(self, layer: int):
        name = C.create_string_buffer(64)
        ci, co, h, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        fl = C.c_double()
        self._ck(lib().aipt_denoise_layer_info(self._h, layer, name, 64, C.byref(ci), C.byref(co), C.byref(h),
                                               C.byref(w), C.byref(fl)))
        return dict(kernel=name.value.decode(), cin=ci.value, cout=co.value, h=h.value, w=w.value, flops=fl.value)

    # ------------------------------------------------------------------ frame (runCuda, main.cpp:143-163)
    def frame_configure(self, width: int, height: int):
        self._ck(lib().aipt_frame_configure(self._h, width, height))

    def frame(self, cam: Camera, iter: int, depth: int, out3, trace_flags: int = TRACE_DEFAULT,
              bn_batch: bool = True, carry: bool = True):
        flags = (DN_BN_BATCH if bn_batch else 0) | (DN_HIDDEN_CARRY if carry else 0)
        self._ck(lib().aipt_frame(self._h, C.byref(cam), iter, depth, trace_flags, flags, _P(out3.data_ptr())))

    def trace_configure_batch(self, width: int, height: int, batch: int):
        self._ck(lib().aipt_trace_configure_batch(self._h, width, height, batch))

    def pathtrace_batch(self, cams, iter: int, depth: int, gbufs, flags: int = TRACE_DEFAULT):
        """aipt_trace_batch: len(cams) frames into gbufs, a float32 device tensor [nframes, 10, rows, stride]"""
        assert gbufs.is_cuda and gbufs.is_contiguous() and gbufs.dim() == 4 and gbufs.shape[0] >= len(cams) and gbufs.shape[1] == 10
        ca = (Camera * len(cams))(*cams)
        self._ck(lib().aipt_trace_batch(self._h, ca, len(cams), iter, depth, flags, _P(gbufs.data_ptr()), gbufs.shape[2],
                                        gbufs.shape[3], 10 * gbufs.shape[2] * gbufs.shape[3]))

    def live_counts_frame(self, frame: int, depth: int) -> np.ndarray:
        out = np.zeros(depth + 1, np.int32)
        self._ck(lib().aipt_trace_live_counts_frame(self._h, frame, out.ctypes.data, depth + 1))
        return out

    def frames_configure(self, batch: int):
        self._ck(lib().aipt_frames_configure(self._h, batch))

    def frames(self, cams, iter: int, depth: int, outs, trace_flags: int = TRACE_DEFAULT, bn_batch: bool = True,
               carry_first: bool = True, carry: bool = True):
        """aipt_frames: trace len(cams) consecutive frames with one set of launches, denoise them in order into outs[j]."""
        ca = (Camera * len(cams))(*cams)
        ptrs = (_P * len(cams))(*[_P(o.data_ptr()) for o in outs[:len(cams)]])
        f0 = (DN_BN_BATCH if bn_batch else 0) | (DN_HIDDEN_CARRY if carry_first else 0)
        f1 = (DN_BN_BATCH if bn_batch else 0) | (DN_HIDDEN_CARRY if carry else 0)
        self._ck(lib().aipt_frames(self._h, ca, len(cams), iter, depth, trace_flags, f0, f1, ptrs))

    def frames_prefetch(self, cams, iter: int, depth: int, trace_flags: int = TRACE_DEFAULT):
        ca = (Camera * len(cams))(*cams)
        self._ck(lib().aipt_frames_prefetch(self._h, ca, len(cams), iter, depth, trace_flags))

    def frames_gbuffer(self, frame: int):
        p, r, s = _P(), C.c_int(), C.c_int()
        self._ck(lib().aipt_frames_gbuffer(self._h, frame, C.byref(p), C.byref(r), C.byref(s)))
        return p.value, r.value, s.value

    def frame_prefetch(self, cam: Camera, iter: int, depth: int, trace_flags: int = TRACE_DEFAULT):
        """Trace the NEXT frame on the side stream while the current one is denoised; the next frame() call with the
        same arguments consumes it."""
        self._ck(lib().aipt_frame_prefetch(self._h, C.byref(cam), iter, depth, trace_flags))

    def gbuffer(self):
        """(device pointer, rows, stride) of the context-owned padded G-buffer."""
        p, r, s = _P(), C.c_int(), C.c_int()
        self._ck(lib().aipt_gbuffer(self._h, C.byref(p), C.byref(r), C.byref(s)))
        return p.value, r.value, s.value

    def frame_set_timing(self, on: bool):
        self._ck(lib().aipt_frame_set_timing(self._h, 1 if on else 0))

    def frame_last_times(self):
        a, b = C.c_float(), C.c_float()
        self._ck(lib().aipt_frame_last_times(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
