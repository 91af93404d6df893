"""ctypes view of the C ABI declared in include/bpvo_hip/c_api.h.

`Binding(lib_path, prefix)` is prefix-agnostic so that the tests can point the very same wrapper at a second library
with the same call shapes (the CPU checker used by tests/) and compare call by call; the product (`bpvo_amd.load()`)
only ever loads libbpvo_hip.so and raises if it is missing — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

LOSS_HUBER, LOSS_TUKEY, LOSS_L2 = 0x10, 0x11, 0x12
VERB_ITERATION, VERB_FINAL, VERB_SILENT, VERB_DEBUG = 0x20, 0x21, 0x22, 0x23
DESC_INTENSITY, DESC_GRADIENT, DESC_LAPLACIAN, DESC_BITPLANES = 0x30, 0x31, 0x36, 0x37
DESC_FIELDS1, DESC_FIELDS2, DESC_LATCH, DESC_CENTRAL_DIFFERENCE = 0x32, 0x33, 0x34, 0x35
GRAD_CD3, GRAD_CD5 = 0, 1
INTERP_LINEAR, INTERP_COSINE, INTERP_CUBIC, INTERP_CUBIC_HERMITE = 0, 1, 2, 3
STATUS_PARAMETER_TOL, STATUS_FUNCTION_TOL, STATUS_GRADIENT_TOL, STATUS_MAX_ITERATIONS, STATUS_SOLVER_ERROR = range(0x30, 0x35)
KF_LARGE_TRANSLATION, KF_LARGE_ROTATION, KF_SMALL_FRAC_GOOD, KF_NO_KEYFRAMING, KF_FIRST_FRAME = range(0x40, 0x45)
MAX_LEVELS = 8
TRACE_FLOATS = 68
STEREO_BM, STEREO_SGM, STEREO_SGBM = 0, 1, 2


class Params(C.Structure):
    """POD mirror of bpvo::AlgorithmParameters (reference: bpvo/types.h:171-413)."""
    _fields_ = [
        ("numPyramidLevels", C.c_int), ("minImageDimensionForPyramid", C.c_int),
        ("sigmaPriorToCensusTransform", C.c_float), ("sigmaBitPlanes", C.c_float),
        ("dfSigma1", C.c_float), ("dfSigma2", C.c_float),
        ("latchNumBytes", C.c_int), ("latchRotationInvariance", C.c_int), ("latchHalfSsdSize", C.c_int),
        ("centralDifferenceRadius", C.c_int),
        ("centralDifferenceSigmaBefore", C.c_float), ("centralDifferenceSigmaAfter", C.c_float),
        ("laplacianKernelSize", C.c_int), ("maxIterations", C.c_int),
        ("parameterTolerance", C.c_float), ("functionTolerance", C.c_float), ("gradientTolerance", C.c_float),
        ("relaxTolerancesForCoarseLevels", C.c_int), ("gradientEstimation", C.c_int), ("interp", C.c_int),
        ("lossFunction", C.c_int), ("descriptor", C.c_int), ("verbosity", C.c_int),
        ("minTranslationMagToKeyFrame", C.c_float), ("minRotationMagToKeyFrame", C.c_float),
        ("maxFractionOfGoodPointsToKeyFrame", C.c_float), ("goodPointThreshold", C.c_float),
        ("minNumPixelsForNonMaximaSuppression", C.c_int), ("nonMaxSuppRadius", C.c_int), ("minNumPixelsToWork", C.c_int),
        ("minSaliency", C.c_float), ("minValidDisparity", C.c_float), ("maxValidDisparity", C.c_float),
        ("maxTestLevel", C.c_int), ("withNormalization", C.c_int),
    ]


class Stats(C.Structure):
    _fields_ = [("numIterations", C.c_int), ("finalError", C.c_float), ("firstOrderOptimality", C.c_float), ("status", C.c_int)]


class Result(C.Structure):
    _fields_ = [("pose", C.c_float * 16), ("covariance", C.c_float * 36), ("optimizerStatistics", Stats * MAX_LEVELS),
                ("numLevels", C.c_int), ("isKeyFrame", C.c_int), ("keyFramingReason", C.c_int), ("hasPointCloud", C.c_int)]


class StereoParams(C.Structure):
    """bpvo_hip_stereo_params: the CvStereoBMState fields the reference sets (utils/stereo_algorithm.cc:63-82)."""
    _fields_ = [("preFilterCap", C.c_int), ("SADWindowSize", C.c_int), ("minDisparity", C.c_int), ("numberOfDisparities", C.c_int),
                ("textureThreshold", C.c_int), ("uniquenessRatio", C.c_int),
                # StereoAlgorithm: 0 block matching, 1 SgmStereo (utils/sgm.h:33-46) with the fields below
                ("algorithm", C.c_int), ("sobelCapValue", C.c_int), ("censusRadius", C.c_int), ("windowRadius", C.c_int),
                ("smoothnessPenaltySmall", C.c_int), ("smoothnessPenaltyLarge", C.c_int), ("consistencyThreshold", C.c_int), ("reserved_", C.c_int),
                ("disparityFactor", C.c_double), ("censusWeightFactor", C.c_double),
                # StereoAlgorithm 2: cv::StereoSGBM (OpenCV 2.4) — its fields; it also reads minDisparity, numberOfDisparities, SADWindowSize,
                # preFilterCap, uniquenessRatio above
                ("P1", C.c_int), ("P2", C.c_int), ("disp12MaxDiff", C.c_int), ("speckleWindowSize", C.c_int), ("speckleRange", C.c_int), ("fullDP", C.c_int)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double), ("units", C.c_double),
                ("bytes_per_unit", C.c_double)]


POINT_WITH_INFO = np.dtype([("xyzw", np.float32, 4), ("rgba", np.uint8, 4), ("weight", np.float32), ("pad", np.uint8, 8)])
assert POINT_WITH_INFO.itemsize == 32


class BpvoError(RuntimeError):
    """Non-zero status from the C ABI (the C++ facade maps the same statuses to bpvo::Error)."""


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Binding:
    def __init__(self, lib_path: str, prefix: str):
        if not os.path.exists(lib_path):
            raise FileNotFoundError(lib_path)
        self.lib = C.CDLL(lib_path)
        self.prefix = prefix
        self.path = lib_path

    def fn(self, name, restype=C.c_int):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        return f

    def has(self, name):
        return hasattr(self.lib, self.prefix + name)

    def default_params(self) -> Params:
        p = Params()
        self.fn("default_params", None)(C.byref(p))
        return p

    def create(self, K, baseline, rows, cols, params: Params, device=0, n_frames=3, n_pairs=1) -> "Context":
        return Context(self, K, baseline, rows, cols, params, device, n_frames, n_pairs)


class Context:
    """One bpvo_*_ctx. Methods mirror the C ABI one to one and return numpy arrays in the reference layouts."""

    def __init__(self, b: Binding, K, baseline, rows, cols, params, device, n_frames, n_pairs):
        self.b = b
        self.h = C.c_void_p()
        self.rows, self.cols = int(rows), int(cols)
        Kf = _f32(K).reshape(9)
        rc = b.fn("create")(C.byref(self.h), Kf.ctypes.data_as(C.c_void_p), C.c_float(baseline), int(rows), int(cols),
                            C.byref(params), int(device), int(n_frames), int(n_pairs))
        if rc != 0:
            msg = b.fn("last_error", C.c_char_p)(None)
            raise BpvoError(f"{b.prefix}create failed ({rc}): {msg.decode() if msg else ''}")
        self.n_frames, self.n_pairs = n_frames, n_pairs
        self.L = b.fn("num_levels")(self.h)
        self.Cn = b.fn("num_channels")(self.h)

    def close(self):
        if self.h:
            self.b.fn("destroy", None)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            msg = self.b.fn("last_error", C.c_char_p)(self.h)
            raise BpvoError(f"status {rc}: {msg.decode() if msg else ''}")

    def call(self, name, *args):
        self._ck(self.b.fn(name)(self.h, *args))

    def set_warp_formulation(self, mode):
        """0 = PhotoError f64 (active reference path, default), 1 = projectPoints / BilinearInterp all-f32 formulation,
        2 = DisparitySpaceWarp as the warp (+ the f32 interpolation); drops the templates when the warp changes."""
        self.call("set_warp_formulation", int(mode))

    # -- geometry
    def level_size(self, level):
        r, c = C.c_int(), C.c_int()
        self.call("level_size", int(level), C.byref(r), C.byref(c))
        return r.value, c.value

    # -- frames
    def frame_set_data(self, slot, img, disp):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        disp = _f32(disp)
        assert img.shape == (self.rows, self.cols) and disp.shape == (self.rows, self.cols)
        self.call("frame_set_data", int(slot), img.ctypes.data_as(C.c_void_p), disp.ctypes.data_as(C.c_void_p))

    def frame_set_template(self, slot):
        self.call("frame_set_template", int(slot))

    def frame_clear(self, slot):
        self.call("frame_clear", int(slot))

    def frame_state(self, slot):
        a, b = C.c_int(), C.c_int()
        self.call("frame_state", int(slot), C.byref(a), C.byref(b))
        return bool(a.value), bool(b.value)

    def frames_set_data(self, first, stride, images, disps):
        images = np.ascontiguousarray(images, dtype=np.uint8)
        disps = _f32(disps)
        self.call("frames_set_data", int(first), int(stride), int(images.shape[0]), images.ctypes.data_as(C.c_void_p),
                  disps.ctypes.data_as(C.c_void_p), 0)

    def frames_set_data_device(self, first, stride, count, d_images_ptr, d_disps_ptr):
        self.call("frames_set_data", int(first), int(stride), int(count), C.c_void_p(d_images_ptr), C.c_void_p(d_disps_ptr), 1)

    def frames_set_template(self, first, stride, count):
        self.call("frames_set_template", int(first), int(stride), int(count))

    # -- accessors
    def get_image(self, slot, level):
        r, c = self.level_size(level)
        out = np.empty((r, c), np.uint8)
        self.call("get_image", int(slot), int(level), out.ctypes.data_as(C.c_void_p))
        return out

    def get_descriptor_channel(self, slot, level, ch):
        r, c = self.level_size(level)
        out = np.empty((r, c), np.float32)
        self.call("get_descriptor_channel", int(slot), int(level), int(ch), out.ctypes.data_as(C.c_void_p))
        return out

    def get_saliency(self, slot, level):
        r, c = self.level_size(level)
        out = np.empty((r, c), np.float32)
        self.call("get_saliency", int(slot), int(level), out.ctypes.data_as(C.c_void_p))
        return out

    def num_points(self, slot, level):
        n = C.c_int()
        self.call("num_points", int(slot), int(level), C.byref(n))
        return n.value

    def get_points(self, slot, level):
        out = np.empty((self.num_points(slot, level), 4), np.float32)
        self.call("get_points", int(slot), int(level), out.ctypes.data_as(C.c_void_p))
        return out

    def get_point_indices(self, slot, level):
        out = np.empty(self.num_points(slot, level), np.int32)
        self.call("get_point_indices", int(slot), int(level), out.ctypes.data_as(C.c_void_p))
        return out

    def get_pixels(self, slot, level):
        out = np.empty((self.Cn, self.num_points(slot, level)), np.float32)
        self.call("get_pixels", int(slot), int(level), out.ctypes.data_as(C.c_void_p))
        return out

    def get_jacobians(self, slot, level):
        out = np.empty((self.Cn, self.num_points(slot, level), 6), np.float32)
        self.call("get_jacobians", int(slot), int(level), out.ctypes.data_as(C.c_void_p))
        return out

    def get_normalization(self, slot, level):
        T, Ti = np.empty((4, 4), np.float32), np.empty((4, 4), np.float32)
        self.call("get_normalization", int(slot), int(level), T.ctypes.data_as(C.c_void_p), Ti.ctypes.data_as(C.c_void_p))
        return T, Ti

    # -- operator-level seam
    def linearize(self, ws, ref, cur, level, T, reset_scale=True):
        T = _f32(T).reshape(16)
        H, G = np.empty((6, 6), np.float32), np.empty(6, np.float32)
        f, s, nv = C.c_float(), C.c_float(), C.c_int()
        self.call("linearize", int(ws), int(ref), int(cur), int(level), T.ctypes.data_as(C.c_void_p), int(bool(reset_scale)),
                  H.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p), C.byref(f), C.byref(s), C.byref(nv))
        return dict(H=H, G=G, f_norm=f.value, sigma=s.value, num_valid=nv.value)

    def linearize_at_scale(self, ws, ref, cur, level, T, sigma):
        """computeResiduals at T, then ComputeWeights / LinearSystemBuilder::Run with the GIVEN robust scale (HIP library only)."""
        T = _f32(T).reshape(16)
        H, G = np.empty((6, 6), np.float32), np.empty(6, np.float32)
        f, nv = C.c_float(), C.c_int()
        self.call("linearize_at_scale", int(ws), int(ref), int(cur), int(level), T.ctypes.data_as(C.c_void_p), C.c_float(sigma),
                  H.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p), C.byref(f), C.byref(nv))
        return dict(H=H, G=G, f_norm=f.value, sigma=float(np.float32(sigma)), num_valid=nv.value)

    def _get_vec(self, name, ws, dtype):
        n = C.c_size_t()
        self.call(name, int(ws), None, C.byref(n))
        out = np.empty(n.value, dtype)
        self.call(name, int(ws), out.ctypes.data_as(C.c_void_p), C.byref(n))
        return out

    def get_residuals(self, ws=0):
        return self._get_vec("get_residuals", ws, np.float32)

    def get_valid(self, ws=0):
        return self._get_vec("get_valid", ws, np.uint16)

    def get_weights(self, ws=0):
        return self._get_vec("get_weights", ws, np.float32)

    def fraction_good(self, ws, thr):
        f = C.c_float()
        self.call("fraction_good", int(ws), C.c_float(thr), C.byref(f))
        return f.value

    # -- estimatePose
    def estimate_pose(self, ws, ref, cur, T_init=None):
        T0 = _f32(np.eye(4) if T_init is None else T_init).reshape(16)
        T = np.empty((4, 4), np.float32)
        st = (Stats * self.L)()
        self.call("estimate_pose", int(ws), int(ref), int(cur), T0.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p), st)
        return T, [dict(numIterations=s.numIterations, finalError=s.finalError,
                        firstOrderOptimality=s.firstOrderOptimality, status=s.status) for s in st]

    def estimate_pose_trace(self, ws, ref, cur, T_init=None, max_records=4096):
        T0 = _f32(np.eye(4) if T_init is None else T_init).reshape(16)
        T = np.empty((4, 4), np.float32)
        st = (Stats * self.L)()
        rec = np.zeros((max_records, TRACE_FLOATS), np.float32)
        n = C.c_int()
        self.call("estimate_pose_trace", int(ws), int(ref), int(cur), T0.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p),
                  st, rec.ctypes.data_as(C.c_void_p), int(max_records), C.byref(n))
        stats = [dict(numIterations=s.numIterations, finalError=s.finalError,
                      firstOrderOptimality=s.firstOrderOptimality, status=s.status) for s in st]
        return T, stats, rec[: min(n.value, max_records)]

    # -- VisualOdometry
    def add_frame(self, img, disp):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        disp = _f32(disp)
        r = Result()
        self.call("add_frame", img.ctypes.data_as(C.c_void_p), disp.ctypes.data_as(C.c_void_p), C.byref(r))
        return dict(pose=np.array(r.pose, np.float32).reshape(4, 4), covariance=np.array(r.covariance, np.float32).reshape(6, 6),
                    stats=[dict(numIterations=s.numIterations, finalError=s.finalError,
                                firstOrderOptimality=s.firstOrderOptimality, status=s.status)
                           for s in r.optimizerStatistics[: r.numLevels]],
                    isKeyFrame=bool(r.isKeyFrame), keyFramingReason=r.keyFramingReason, hasPointCloud=bool(r.hasPointCloud))

    def default_stereo_params(self, ndisp=64):
        sp = StereoParams()
        self.b.fn("default_stereo_params", None)(C.byref(sp))
        sp.numberOfDisparities = int(ndisp)
        return sp

    def sgbm_params_from_config(self, minDisparity, numberOfDisparities, SADWindowSize=3, P1=0, P2=0, uniquenessRatio=0, speckleWindowSize=0,
                                speckleRange=0, fullDP=0):
        """bpvo_hip_stereo_params_sgbm_from_config: the struct the reference's StereoSGBM constructor call builds from the config KEYS."""
        sp = StereoParams()
        self.b.fn("stereo_params_sgbm_from_config", None)(C.byref(sp), int(minDisparity), int(numberOfDisparities), int(SADWindowSize), int(P1), int(P2),
                                                          int(uniquenessRatio), int(speckleWindowSize), int(speckleRange), int(fullDP))
        return sp

    def stereo_bm(self, left, right, sp):
        """StereoAlgorithm::run (block matching) on one pair or a stack [n, rows, cols]: f32 disparities, invalid = minDisparity - 1."""
        left = np.ascontiguousarray(left, dtype=np.uint8)
        right = np.ascontiguousarray(right, dtype=np.uint8)
        n = 1 if left.ndim == 2 else left.shape[0]
        out = np.empty(left.shape, np.float32)
        self.call("stereo_bm", n, left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p), 0, C.byref(sp), out.ctypes.data_as(C.c_void_p), 0)
        return out

    def stereo_bm_device(self, count, d_left_ptr, d_right_ptr, sp, d_disp_ptr):
        """StereoAlgorithm::run on `count` pairs already in device memory, f32 disparities written to device memory."""
        self.call("stereo_bm", int(count), C.c_void_p(d_left_ptr), C.c_void_p(d_right_ptr), 1, C.byref(sp), C.c_void_p(d_disp_ptr), 1)

    def add_frame_stereo(self, left, right, sp):
        left = np.ascontiguousarray(left, dtype=np.uint8)
        right = np.ascontiguousarray(right, dtype=np.uint8)
        r = Result()
        self.call("add_frame_stereo", left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p), C.byref(sp), C.byref(r))
        return dict(pose=np.array(r.pose, np.float32).reshape(4, 4), stats=[dict(numIterations=s.numIterations, finalError=s.finalError,
                    firstOrderOptimality=s.firstOrderOptimality, status=s.status) for s in r.optimizerStatistics[: r.numLevels]],
                    isKeyFrame=bool(r.isKeyFrame), keyFramingReason=r.keyFramingReason, hasPointCloud=bool(r.hasPointCloud))

    def add_frame_null(self):
        """addFrame(nullptr, nullptr) — must fail like THROW_ERROR_IF at bpvo/vo.cc:68-69."""
        r = Result()
        return self.b.fn("add_frame")(self.h, None, None, C.byref(r))

    def vo_num_points_at_level(self, level=-1):
        n = C.c_int()
        self.call("vo_num_points_at_level", int(level), C.byref(n))
        return n.value

    def vo_points_at_level(self, level=-1):
        out = np.empty((self.vo_num_points_at_level(level), 4), np.float32)
        self.call("vo_points_at_level", int(level), out.ctypes.data_as(C.c_void_p))
        return out

    def get_point_cloud(self):
        n = C.c_size_t()
        pose = np.empty((4, 4), np.float32)
        self.call("get_point_cloud", None, C.byref(n), pose.ctypes.data_as(C.c_void_p))
        pts = np.zeros(n.value, POINT_WITH_INFO)
        self.call("get_point_cloud", pts.ctypes.data_as(C.c_void_p), C.byref(n), pose.ctypes.data_as(C.c_void_p))
        return pts, pose

    def trajectory(self):
        n = C.c_int()
        self.call("trajectory_size", C.byref(n))
        out = np.empty((n.value, 4, 4), np.float32)
        if n.value:
            self.call("get_trajectory", out.ctypes.data_as(C.c_void_p))
        return out

    # -- batches
    def _stats_array(self, st, n_pairs):
        a = np.ctypeslib.as_array(C.cast(st, C.POINTER(C.c_int32)), shape=(n_pairs, self.L, 4)).copy()
        out = np.empty((n_pairs, self.L), dtype=[("numIterations", np.int32), ("finalError", np.float32),
                                                  ("firstOrderOptimality", np.float32), ("status", np.int32)])
        out["numIterations"] = a[..., 0]
        out["finalError"] = a[..., 1].view(np.float32)
        out["firstOrderOptimality"] = a[..., 2].view(np.float32)
        out["status"] = a[..., 3]
        return out

    def batch_run(self, images, disps):
        images = np.ascontiguousarray(images, dtype=np.uint8)
        disps = _f32(disps)
        n_pairs = images.shape[0] // 2
        poses = np.empty((n_pairs, 4, 4), np.float32)
        st = (Stats * (n_pairs * self.L))()
        self.call("batch_run", n_pairs, images.ctypes.data_as(C.c_void_p), disps.ctypes.data_as(C.c_void_p), 0,
                  poses.ctypes.data_as(C.c_void_p), st)
        return poses, self._stats_array(st, n_pairs)

    def batch_run_device(self, n_pairs, d_images_ptr, d_disps_ptr):
        poses = np.empty((n_pairs, 4, 4), np.float32)
        st = (Stats * (n_pairs * self.L))()
        self.call("batch_run", int(n_pairs), C.c_void_p(d_images_ptr), C.c_void_p(d_disps_ptr), 1,
                  poses.ctypes.data_as(C.c_void_p), st)
        return poses, self._stats_array(st, n_pairs)

    def batch_estimate(self, n_pairs, T_init=None):
        poses = np.empty((n_pairs, 4, 4), np.float32)
        st = (Stats * (n_pairs * self.L))()
        T0 = None if T_init is None else _f32(T_init).reshape(-1)
        self.call("batch_estimate", int(n_pairs), None if T0 is None else T0.ctypes.data_as(C.c_void_p),
                  poses.ctypes.data_as(C.c_void_p), st)
        return poses, self._stats_array(st, n_pairs)

    def batch_result_records_device(self):
        p, n = C.c_void_p(), C.c_int()
        self.call("batch_result_records_device", C.byref(p), C.byref(n))
        return p.value, n.value

    def batch_copy_records_device(self, d_dst_ptr, n_pairs):
        self.call("batch_copy_records_device", C.c_void_p(d_dst_ptr), int(n_pairs))

    # -- measurement
    def profiling(self, level=1):
        """0 off, 1 = HIP events around the frame stages + every 5th warp_residual launch, 2 = around every kernel, 3 = as 1 with every
        warp_residual launch. Resets the counters."""
        self.call("profiling", int(level))

    def kernel_stats(self):
        arr = (KernelStat * 32)()
        n = C.c_int()
        self.call("get_kernel_stats", arr, 32, C.byref(n))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms, units=arr[i].units,
                     bytes_per_unit=arr[i].bytes_per_unit) for i in range(n.value)]

    def median_path_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.call("median_path_counts", C.byref(a), C.byref(b))
        return a.value, b.value

    def fused_point_counts(self):
        """(fused, total) points linearised since the last counter reset (HIP library only)."""
        a, b = C.c_uint64(), C.c_uint64()
        self.call("fused_point_counts", C.byref(a), C.byref(b))
        return a.value, b.value

    def tap_cache_counts(self):
        """(hits, lookups, hits in the first 8 linearisations of a level, lookups there) since the last counter reset."""
        a = (C.c_uint64 * 4)()
        self.call("tap_cache_counts", a)
        return tuple(int(x) for x in a)

    def persistent_counts(self):
        """(pyramid levels run by the persistent single-launch GN kernel, 1 if such a launch ever gave up) (HIP library only)."""
        a, b = C.c_uint64(), C.c_int()
        self.call("persistent_counts", C.byref(a), C.byref(b))
        return a.value, b.value

    def upload_stats(self):
        """(seconds, bytes) of the upload pipeline of the last host-buffer batch_run (HIP library only)."""
        a, b = C.c_double(), C.c_uint64()
        self.call("upload_stats", C.byref(a), C.byref(b))
        return a.value, b.value

    def team_counts(self):
        """Batch estimates run by the team-persistent kernel since the context was created (HIP library only)."""
        a = C.c_uint64()
        self.call("team_counts", C.byref(a))
        return a.value

    def set_option(self, key, value):
        """bpvo_hip_set_option: per-context scheduling options (include/bpvo_hip/c_api.h)."""
        self.b.fn("set_option").argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        self._ck(self.b.fn("set_option")(self.h, key.encode(), C.c_double(float(value))))

    def get_option(self, key):
        v = C.c_double(0.0)
        self.b.fn("get_option").argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
        self._ck(self.b.fn("get_option")(self.h, key.encode(), C.byref(v)))
        return v.value

    def set_max_lanes(self, n):
        """Cap (n >= 1) or uncap (n <= 0) the estimation lanes of later batch calls (HIP library only)."""
        self.call("set_max_lanes", int(n))

    def total_linearizations(self):
        n = C.c_uint64()
        self.call("total_linearizations", C.byref(n))
        return n.value
