"""plviwo_amd — Python plumbing over the C-ABI of the MI355X-native PL-VIWO hot path.

The product is the gfx950 shared library ``lib/libplviwo_hip.so`` (sources under ``csrc/``,
boundary in ``include/plviwo.h``).  This module only binds that C-ABI with ctypes so tests,
``bench.py`` and ``__graft_entry__`` can drive it; there is no Python or CPU compute path here,
and loading fails loudly when the library is missing.

The directory is named ``pl-viwo_amd`` (not importable as written); load it with
``tests/conftest.py::load_pkg`` / ``__graft_entry__.load_pkg`` which registers it as
``plviwo_amd``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libplviwo_hip.so")

PLV_OK = 0
PLV_E_BADARG = -1
PLV_E_DEVICE = -2
PLV_E_NOT_PSD = -3
PLV_E_NOMEM = -4
PLV_E_CAPACITY = -5
PLV_E_NO_DEVICE = -6
PLV_E_NUMERIC = -7


class PlvConfig(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("num_features", C.c_int), ("fast_threshold", C.c_int),
        ("grid_x", C.c_int), ("grid_y", C.c_int), ("min_px_dist", C.c_int),
        ("histogram_method", C.c_int), ("win_size", C.c_int), ("pyr_levels", C.c_int),
        ("lk_max_iters", C.c_int), ("lk_eps", C.c_float),
        ("ransac_thr_px", C.c_double), ("ransac_conf", C.c_double), ("ransac_max_iters", C.c_int),
        ("intrinsics", C.c_double * 8),
        ("line_length_threshold", C.c_int), ("line_distance_threshold", C.c_float),
        ("canny_th1", C.c_int), ("canny_th2", C.c_int), ("canny_aperture", C.c_int),
        ("line_min_length_px", C.c_float), ("line_assign_px", C.c_float), ("line_similar_px", C.c_float),
        ("max_state_dim", C.c_int), ("max_meas_rows", C.c_int), ("max_features", C.c_int),
        ("max_rows_per_feat", C.c_int),
        ("sigma_pix", C.c_double), ("chi2_mult", C.c_double),
        ("device", C.c_int),
    ]


class PlvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"plv error {code}: {msg}")
        self.code = code


_lib = None


def load_library():
    """Loads libplviwo_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    dp, ip, u8p = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint8)
    fp = C.POINTER(C.c_float)
    u64p = C.POINTER(C.c_uint64)
    vp = C.c_void_p
    sig = {
        "plv_abi_version": (C.c_int, []),
        "plv_last_error": (C.c_char_p, []),
        "plv_device_count": (C.c_int, []),
        "plv_device_numa_node": (C.c_int, [C.c_int]),
        "plv_config_default": (None, [C.POINTER(PlvConfig), C.c_int, C.c_int]),
        "plv_ctx_create": (C.c_int, [C.POINTER(PlvConfig), C.POINTER(vp)]),
        "plv_ctx_destroy": (None, [vp]),
        "plv_ctx_synchronize": (C.c_int, [vp]),
        "plv_prof_enable": (C.c_int, [vp, C.c_int]),
        "plv_prof_reset": (C.c_int, [vp]),
        "plv_prof_count": (C.c_int, [vp]),
        "plv_prof_get": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_int, ip, dp]),
        "plv_ekf_update": (C.c_int, [vp, dp, C.c_int, C.c_int, dp, C.c_int, C.c_int, C.c_int, ip, dp, dp, dp]),
        "plv_cov_upload": (C.c_int, [vp, dp, C.c_int, C.c_int]),
        "plv_cov_download": (C.c_int, [vp, dp, C.c_int, C.c_int]),
        "plv_cov_checkpoint": (C.c_int, [vp]),
        "plv_cov_rollback": (C.c_int, [vp]),
        "plv_compress": (C.c_int, [vp, dp, C.c_int, C.c_int, C.c_int, dp, ip]),
        "plv_nullspace_batch": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp, dp, dp]),
        "plv_chi2_batch": (C.c_int, [vp, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp, dp, ip,
                                     C.c_double, dp]),
        "plv_chi2_quantile95": (C.c_double, [C.c_int]),
        "plv_msckf_update": (C.c_int, [vp, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp, dp, dp,
                                       ip, C.c_double, C.c_double, C.c_double, u8p, ip, dp]),
        "plv_feat_batch_upload": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp, dp, dp, ip]),
        "plv_msckf_update_resident": (C.c_int, [vp, C.c_double, C.c_double, C.c_double, u8p, ip, dp]),
        "plv_msckf_update_resident_launch": (C.c_int, [vp, C.c_double, C.c_double, C.c_double]),
        "plv_msckf_update_resident_wait": (C.c_int, [vp, u8p, ip, dp]),
        "plv_feed_image": (C.c_int, [vp, u8p, C.c_int]),
        "plv_image_stage": (C.c_int, [vp, C.c_int, u8p, C.c_int]),
        "plv_feed_staged": (C.c_int, [vp, C.c_int]),
        "plv_image_buffer": (C.c_int, [vp, C.c_int, C.POINTER(u8p), ip]),
        "plv_pyramid_levels": (C.c_int, [vp, C.c_int]),
        "plv_pyramid_download": (C.c_int, [vp, C.c_int, C.c_int, ip, ip, u8p]),
        "plv_lk_track": (C.c_int, [vp, C.c_int, fp, fp, u8p, ip]),
        "plv_undistort": (C.c_int, [vp, C.c_int, fp, fp]),
        "plv_ransac_fundamental": (C.c_int, [vp, C.c_int, fp, fp, C.c_double, C.c_uint32, u8p, ip, ip]),
        "plv_perform_matching": (C.c_int, [vp, C.c_int, fp, fp, u8p, fp, fp, C.POINTER(C.c_longlong)]),
        "plv_update_graph_mode": (C.c_int, [vp, C.c_int, ip, ip]),
        "plv_perform_matching_launch": (C.c_int, [vp, C.c_int, fp, fp]),
        "plv_perform_matching_wait": (C.c_int, [vp, fp, u8p, fp, fp, C.POINTER(C.c_longlong)]),
        "plv_perform_detection": (C.c_int, [vp, C.c_int, u8p, fp, C.POINTER(C.c_uint64), C.c_int, C.c_int,
                                            C.POINTER(C.c_uint64), ip]),
        "plv_tracker_feed": (C.c_int, [vp, C.c_double, u8p, C.c_int, u8p]),
        "plv_tracker_feed_staged": (C.c_int, [vp, C.c_double, C.c_int, u8p]),
        "plv_tracker_detect_ahead": (C.c_int, [vp, C.c_int]),
        "plv_tracker_feed_downsampled": (C.c_int, [vp, C.c_double, u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int]),
        "plv_downsample": (C.c_int, [vp, u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int]),
        "plv_feed_image_downsampled": (C.c_int, [vp, u8p, C.c_int, C.c_int, C.c_int]),
        "plv_tracker_last": (C.c_int, [vp, fp, C.POINTER(C.c_uint64), C.c_int, ip]),
        "plv_db_size": (C.c_int, [vp]),
        "plv_db_select": (C.c_int, [vp, C.c_int, C.c_double, C.POINTER(C.c_uint64), C.c_int, ip]),
        "plv_db_export_tracks": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int, ip, dp, fp, fp, C.c_int]),
        "plv_db_cleanup_measurements": (C.c_int, [vp, C.c_double]),
        "plv_db_remove": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int]),
        "plv_cpi_noise": (C.c_int, [C.POINTER(PlvStateView), C.POINTER(PlvCpiTable), C.c_int, dp, dp, ip, u8p]),
        "plv_triangulate": (C.c_int, [vp, C.POINTER(PlvStateView), C.POINTER(PlvTracks), C.POINTER(PlvTriOptions), dp, u8p, dp]),
        "plv_jacobian_columns": (C.c_int, [C.POINTER(PlvStateView), C.POINTER(PlvTracks), ip, C.c_int, ip]),
        "plv_build_jacobians": (C.c_int, [vp, C.POINTER(PlvStateView), C.POINTER(PlvTracks), C.c_int, ip, C.c_int, ip, dp,
                                          dp, dp]),
        "plv_build_jacobians_resident": (C.c_int, [vp, C.POINTER(PlvStateView), C.POINTER(PlvTracks), C.c_int, ip, C.c_int]),
        "plv_db_append_measurements": (C.c_int, [vp, C.c_uint64, C.c_int, dp, fp, fp]),
        "plv_camera_update_points": (C.c_int, [vp, C.POINTER(PlvStateView), C.POINTER(PlvUpdateOptions), dp,
                                               C.POINTER(PlvUpdateResult), u64p, u8p, dp]),
        "plv_line_db_append_measurements": (C.c_int, [vp, C.c_uint64, C.c_int, dp, fp, fp, C.c_int, ip, C.c_int]),
        "plv_point_used_insert": (C.c_int, [vp, C.c_uint64, dp, C.c_double]),
        "plv_select_imu_readings": (C.c_int, [C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_int, dp, dp, dp, ip, ip]),
        "plv_reset_cpi": (None, [C.POINTER(PlvCpiAccum), C.POINTER(PlvImuState), C.c_double]),
        "plv_propagate": (C.c_int, [vp, C.POINTER(PlvImuState), C.POINTER(PlvImuNoise), C.c_int, dp, dp, dp, C.POINTER(PlvCpiAccum),
                                    C.POINTER(PlvCpiRecord), C.c_int, C.c_int, dp, dp]),
        "plv_cov_clone": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "plv_cpi_integrate": (C.c_int, [vp, C.POINTER(PlvImuNoise), C.c_double, C.c_double, dp, dp, dp, dp, C.c_int, dp, dp, dp,
                                        C.POINTER(PlvCpiRecord), ip]),
        "plv_select_wheel_data": (C.c_int, [C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_int, dp, dp, dp, ip, ip]),
        "plv_wheel_linear_system": (C.c_int, [vp, C.POINTER(PlvWheelOptions), C.POINTER(PlvWheelState), C.c_int, dp, dp, dp, dp, dp, dp, ip, ip,
                                              ip, dp, dp]),
        "plv_init_imu_static": (C.c_int, [C.c_int, dp, dp, dp, C.c_double, C.c_double, dp, dp, ip]),
        "plv_iw_init_reset": (None, [C.POINTER(PlvIwInitState)]),
        "plv_init_imu_wheel": (C.c_int, [C.POINTER(PlvIwInitOptions), C.POINTER(PlvIwInitState), C.c_int, dp, dp, dp, C.c_int, dp, dp, dp,
                                         dp, ip, ip, dp]),
        "plv_set_camera_intrinsics": (C.c_int, [vp, dp]),
        "plv_set_lk_window": (C.c_int, [vp, C.c_int]),
        "plv_wheel_update": (C.c_int, [vp, C.POINTER(PlvWheelOptions), C.POINTER(PlvWheelState), C.c_int, dp, dp, dp, u8p, dp]),
        "plv_next_clone_time": (C.c_int, [C.POINTER(PlvCloneSchedule), dp, ip]),
        "plv_closest_clone_time": (C.c_int, [C.POINTER(PlvStateView), C.c_int, C.c_double, dp, ip]),
        "plv_traj_header": (C.c_int, [C.c_char_p, C.c_int]),
        "plv_traj_format": (C.c_int, [C.c_char_p, C.c_int, C.c_double, dp, dp, dp]),
        "plv_traj_load": (C.c_int, [C.c_char_p, C.c_int, dp, dp, dp, dp, ip, ip]),
        "plv_traj_length": (C.c_double, [C.c_int, dp]),
        "plv_traj_associate": (C.c_int, [C.c_double, C.c_double, C.c_int, dp, C.c_int, dp, ip, ip, ip]),
        "plv_traj_ate": (C.c_int, [vp, C.c_int, C.c_int, dp, dp, C.c_int, dp, dp, dp, dp, dp, dp, C.POINTER(PlvStats),
                                   C.POINTER(PlvStats)]),
        "plv_cpi_poses": (C.c_int, [vp, C.POINTER(PlvStateView), C.POINTER(PlvCpiTable), C.c_int, dp, dp, dp, u8p]),
        "plv_camera_update_list": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, ip, u64p, ip, dp, fp, fp, dp]),
        "plv_slam_marg_flags": (C.c_int, [vp, C.c_int, u64p, ip, u8p]),
        "plv_camera_get_line_features": (C.c_int, [vp, C.POINTER(PlvStateView)]),
        "plv_update_compression_mode": (C.c_int, [vp, C.c_int, ip, ip]),
        "plv_line_worker_config": (C.c_int, [C.c_int, C.c_int, ip, ip]),
        "plv_debug_knobs": (C.c_uint, [C.c_longlong]),
        "plv_chain_count": (C.c_ulonglong, []),
        "plv_decision_trace": (C.c_int, [C.c_void_p, C.c_int]),
        "plv_last_point_decisions": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]),
        "plv_last_line_decisions": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]),
        "plv_route_counts": (None, [C.POINTER(C.c_ulonglong)]),
        "plv_speculation_counts": (None, [C.POINTER(C.c_ulonglong)]),
        "plv_memory_bytes": (None, [C.POINTER(C.c_ulonglong)]),
        "plv_memory_policy": (C.c_int, [C.c_int, C.c_int, C.c_int]),
        "plv_alloc_count": (C.c_ulonglong, []),
        "plv_phase_counters": (None, [C.POINTER(C.c_ulonglong)]),
        "plv_camera_update_lines": (C.c_int, [vp, C.POINTER(PlvStateView), C.POINTER(PlvUpdateOptions), dp,
                                              C.POINTER(PlvUpdateResult), u64p, u8p, dp, C.c_int]),
        "plv_slam_update": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, dp, dp, ip, C.c_double, u8p, dp]),
        "plv_slam_initialize": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, dp, dp, dp, ip, C.c_double, u8p, dp, dp]),
        "plv_cov_marginalize": (C.c_int, [vp, C.c_int, C.c_int]),
        "plv_detect_lines": (C.c_int, [vp, C.c_int, fp, C.c_int, ip]),
        "plv_line_detect_launch": (C.c_int, [vp, C.c_int]),
        "plv_line_detect_finish": (C.c_int, [vp, C.c_int]),
        "plv_counters": (None, [C.POINTER(C.c_ulonglong)]),
        "plv_jpl_left_update": (None, [C.c_int, dp, dp, dp]),
        "plv_state_boxplus": (C.c_int, [C.c_int, C.POINTER(PlvStateVar), dp, C.c_int]),
        "plv_camera_try_update": (C.c_int, [vp, C.POINTER(PlvStateView), C.POINTER(PlvTryUpdate)]),
        "plv_camera_frame": (C.c_int, [vp, C.POINTER(PlvStateView), C.POINTER(PlvCameraFrameIo)]),
        "plv_line_walk_mode": (C.c_int, [vp, C.c_int]),
        "plv_line_prefetch_mode": (C.c_int, [vp, C.c_int]),
        "plv_line_tracker_feed_async": (C.c_int, [vp, C.c_double, dp]),
        "plv_line_tracker_feed_wait": (C.c_int, [vp]),
        "plv_assign_points_to_lines": (C.c_int, [fp, C.c_int, fp, u64p, C.c_int, ip, ip, u64p, dp, ip, fp, ip]),
        "plv_line_match": (C.c_int, [fp, C.c_int, ip, u64p, fp, C.c_int, ip, u64p, ip]),
        "plv_line_classification": (C.c_int, [fp, dp]),
        "plv_vanishing_points": (C.c_int, [dp, dp, dp]),
        "plv_line_tracker_feed": (C.c_int, [vp, C.c_double, dp]),
        "plv_line_tracker_feed_points": (C.c_int, [vp, C.c_double, dp, C.c_int, fp, u64p]),
        "plv_line_tracker_last": (C.c_int, [vp, fp, u64p, C.c_int, ip]),
        "plv_line_db_size": (C.c_int, [vp]),
        "plv_line_db_ids": (C.c_int, [vp, u64p, C.c_int, ip]),
        "plv_line_db_export_tracks": (C.c_int, [vp, u64p, C.c_int, ip, dp, fp, fp, C.c_int, ip, ip, ip, C.c_int]),
        "plv_line_db_remove": (C.c_int, [vp, u64p, C.c_int]),
        "plv_triangulate_lines": (C.c_int, [vp, C.POINTER(PlvStateView), C.POINTER(PlvLineTracks), dp, u8p]),
        "plv_line_jacobian_columns": (C.c_int, [C.POINTER(PlvStateView), C.POINTER(PlvLineTracks), ip, C.c_int, ip]),
        "plv_build_line_jacobians": (C.c_int, [vp, C.POINTER(PlvStateView), C.POINTER(PlvLineTracks), C.c_int, ip, C.c_int, ip,
                                               dp, dp, dp]),
        "plv_build_line_jacobians_resident": (C.c_int, [vp, C.POINTER(PlvStateView), C.POINTER(PlvLineTracks), C.c_int, ip,
                                                        C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib._plv_signatures = sig
    _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64)) if a is not None else None


def _f64(a):
    return np.asfortranarray(a, dtype=np.float64)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)



class PlvStateView(C.Structure):
    _fields_ = [
        ("n_clones", C.c_int),
        ("clone_time", C.POINTER(C.c_double)), ("clone_R", C.POINTER(C.c_double)), ("clone_p", C.POINTER(C.c_double)),
        ("clone_R_fej", C.POINTER(C.c_double)), ("clone_p_fej", C.POINTER(C.c_double)),
        ("clone_state_id", C.POINTER(C.c_int)),
        ("R_ItoC", C.c_double * 9), ("p_IinC", C.c_double * 3), ("intrinsics", C.c_double * 8), ("cam_dt", C.c_double),
        ("extrinsic_state_id", C.c_int), ("intrinsic_state_id", C.c_int), ("dt_state_id", C.c_int),
        ("intr_order", C.c_int), ("dt_exp", C.c_double), ("sigma_pix", C.c_double),
        ("use_pol_cov", C.c_int), ("intr_ori_cov", C.c_double), ("intr_pos_cov", C.c_double),
        ("feat_rep", C.c_int), ("use_imu_cov", C.c_int), ("intr_err_mlt", C.c_double),
    ]


class PlvTracks(C.Structure):
    _fields_ = [
        ("n_feat", C.c_int),
        ("obs_ptr", C.POINTER(C.c_int)), ("obs_time", C.POINTER(C.c_double)), ("obs_uv", C.POINTER(C.c_float)),
        ("p_FinG", C.POINTER(C.c_double)), ("p_FinG_fej", C.POINTER(C.c_double)),
        ("res_R", C.POINTER(C.c_double)), ("res_p", C.POINTER(C.c_double)),
        ("obs_uvn", C.POINTER(C.c_float)),
        ("res_Q", C.POINTER(C.c_double)), ("res_clone", C.POINTER(C.c_int)),
    ]


class PlvLineTracks(C.Structure):
    _fields_ = [
        ("n_lines", C.c_int),
        ("obs_ptr", C.POINTER(C.c_int)), ("obs_time", C.POINTER(C.c_double)),
        ("seg_uv", C.POINTER(C.c_float)), ("seg_uvn", C.POINTER(C.c_float)),
        ("line_FinG", C.POINTER(C.c_double)), ("D", C.POINTER(C.c_int)),
        ("anchor_pt", C.POINTER(C.c_double)), ("has_pt", C.POINTER(C.c_uint8)),
        ("res_R", C.POINTER(C.c_double)), ("res_p", C.POINTER(C.c_double)),
        ("res_Q", C.POINTER(C.c_double)), ("res_clone", C.POINTER(C.c_int)),
    ]


class PlvTriOptions(C.Structure):
    _fields_ = [("min_dist", C.c_double), ("max_dist", C.c_double), ("max_cond_number", C.c_double),
                ("max_baseline", C.c_double), ("refine_features", C.c_int)]


class PlvStateVar(C.Structure):
    _fields_ = [("kind", C.c_int), ("id", C.c_int), ("size", C.c_int), ("val", C.c_void_p), ("out", C.c_void_p), ("mirror", C.c_void_p)]


class PlvTryUpdate(C.Structure):
    _fields_ = [("opt_points", C.c_void_p), ("opt_lines", C.c_void_p), ("n_var", C.c_int), ("vars", C.c_void_p),
                ("dx_points", C.c_void_p), ("dx_lines", C.c_void_p), ("res_points", C.c_void_p), ("res_lines", C.c_void_p),
                ("msckf_ids", C.c_void_p), ("msckf_accepted", C.c_void_p), ("p_FinG", C.c_void_p),
                ("line_ids", C.c_void_p), ("line_accepted", C.c_void_p), ("line_FinG", C.c_void_p), ("line_cap", C.c_int),
                ("line_db_size", C.c_int)]


class PlvCameraFrameIo(C.Structure):
    _fields_ = [("timestamp", C.c_double), ("slot", C.c_int), ("img", C.c_void_p), ("stride", C.c_int), ("mask", C.c_void_p),
                ("use_lines", C.c_int), ("update", C.c_void_p), ("line_db_size", C.c_int)]


class BoxPlus:
    """A prepared plv_state_boxplus call: the list of variables (arrays updated in place) is laid out once, apply(dx) is one C call.
    entries: (kind 'vec'|'quat', id, value array, out array or None, mirror address or array or None)."""

    def __init__(self, entries):
        self.lib = load_library()
        self.keep = entries            # the arrays must outlive the plan
        self.vars = (PlvStateVar * max(1, len(entries)))()
        self.n = len(entries)
        addr = lambda a: None if a is None else (a if isinstance(a, int) else a.ctypes.data)
        for v, (kind, vid, val, out, mirror) in zip(self.vars, entries):
            assert val.dtype == np.float64 and val.flags.c_contiguous and (out is None or out.flags.c_contiguous)
            v.kind, v.id, v.size = (1 if kind == "quat" else 0), int(vid), (4 if kind == "quat" else val.size)
            v.val, v.out, v.mirror = addr(val), addr(out), addr(mirror)

    def apply(self, dx):
        rc = self.lib.plv_state_boxplus(self.n, self.vars, dx.ctypes.data_as(C.POINTER(C.c_double)), dx.size)
        if rc != 0:
            raise PlvError(rc, "plv_state_boxplus")


class PlvCpiTable(C.Structure):
    _fields_ = [("n", C.c_int), ("t", C.POINTER(C.c_double)), ("clone_t", C.POINTER(C.c_double)), ("dt", C.POINTER(C.c_double)),
                ("R_I0toIk", C.POINTER(C.c_double)), ("alpha", C.POINTER(C.c_double)), ("v", C.POINTER(C.c_double)),
                ("gravity", C.c_double * 3), ("Q", C.POINTER(C.c_double))]


class CpiTable:
    """Owns the arrays behind a plv_cpi_table (State::cpis sorted by time)."""

    def __init__(self, t, clone_t, R, alpha, v, gravity=(0.0, 0.0, 9.81), Q=None):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.t, self.clone_t = f(t), f(clone_t)
        self.Q = f(Q).reshape(-1, 36) if Q is not None else None
        self.dt = self.t - self.clone_t
        self.R, self.alpha, self.v = f(R).reshape(-1, 9), f(alpha).reshape(-1, 3), f(v).reshape(-1, 3)
        c = PlvCpiTable()
        c.n = len(self.t)
        c.t, c.clone_t, c.dt, c.R_I0toIk, c.alpha, c.v = _dp(self.t), _dp(self.clone_t), _dp(self.dt), _dp(self.R), _dp(self.alpha), _dp(self.v)
        c.Q = _dp(self.Q)
        c.gravity = (C.c_double * 3)(*gravity)
        self.c = c


class PlvImuState(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("p", C.c_double * 3), ("v", C.c_double * 3), ("bg", C.c_double * 3), ("ba", C.c_double * 3),
                ("q_fej", C.c_double * 4), ("p_fej", C.c_double * 3), ("v_fej", C.c_double * 3)]

    @classmethod
    def make(cls, q, p, v, bg=(0, 0, 0), ba=(0, 0, 0), q_fej=None, p_fej=None, v_fej=None):
        s = cls()
        for name, val in (("q", q), ("p", p), ("v", v), ("bg", bg), ("ba", ba), ("q_fej", q if q_fej is None else q_fej),
                          ("p_fej", p if p_fej is None else p_fej), ("v_fej", v if v_fej is None else v_fej)):
            arr = getattr(s, name)
            for i, x in enumerate(val):
                arr[i] = float(x)
        return s

    def copy(self):
        c = PlvImuState()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(self))
        return c

    def vec(self):
        return np.array(list(self.q) + list(self.p) + list(self.v) + list(self.bg) + list(self.ba))


class PlvImuNoise(C.Structure):
    _fields_ = [("sigma_w", C.c_double), ("sigma_wb", C.c_double), ("sigma_a", C.c_double), ("sigma_ab", C.c_double),
                ("gravity", C.c_double * 3)]


class PlvCpiAccum(C.Structure):
    _fields_ = [("clone_t", C.c_double), ("DT", C.c_double), ("R_k2tau", C.c_double * 9), ("alpha_tau", C.c_double * 3),
                ("beta_tau", C.c_double * 3), ("b_w_lin", C.c_double * 3), ("b_a_lin", C.c_double * 3), ("v_clone", C.c_double * 3),
                ("P_meas", C.c_double * 225)]

    def copy(self):
        c = PlvCpiAccum()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(self))
        return c


class PlvCpiRecord(C.Structure):
    _fields_ = [("t", C.c_double), ("dt", C.c_double), ("clone_t", C.c_double), ("R_I0toIk", C.c_double * 9), ("alpha", C.c_double * 3),
                ("v", C.c_double * 3), ("w", C.c_double * 3), ("Q", C.c_double * 36)]


def imu_noise(sigma_w=1.6968e-4, sigma_wb=1.9393e-5, sigma_a=2.0e-3, sigma_ab=3.0e-3, gravity=(0.0, 0.0, 9.81)):
    return PlvImuNoise(sigma_w, sigma_wb, sigma_a, sigma_ab, (C.c_double * 3)(*gravity))


class PlvWheelOptions(C.Structure):
    _fields_ = [("type", C.c_int), ("noise_w", C.c_double), ("noise_v", C.c_double), ("noise_p", C.c_double), ("do_calib_ext", C.c_int),
                ("do_calib_dt", C.c_int), ("do_calib_int", C.c_int), ("chi2_mult", C.c_double)]


class PlvWheelState(C.Structure):
    _fields_ = [("intr", C.c_double * 3), ("R_ItoO", C.c_double * 9), ("p_IinO", C.c_double * 3),
                ("R0", C.c_double * 9), ("p0", C.c_double * 3), ("R0_fej", C.c_double * 9), ("p0_fej", C.c_double * 3),
                ("R1", C.c_double * 9), ("p1", C.c_double * 3), ("R1_fej", C.c_double * 9), ("p1_fej", C.c_double * 3),
                ("w0", C.c_double * 3), ("v0", C.c_double * 3), ("w1", C.c_double * 3), ("v1", C.c_double * 3),
                ("pose0_id", C.c_int), ("pose1_id", C.c_int), ("ext_id", C.c_int), ("dt_id", C.c_int), ("intr_id", C.c_int)]

    @classmethod
    def make(cls, intr, R_ItoO, p_IinO, R0, p0, R1, p1, pose0_id, pose1_id, R0_fej=None, p0_fej=None, R1_fej=None, p1_fej=None,
             w0=(0, 0, 0), v0=(0, 0, 0), w1=(0, 0, 0), v1=(0, 0, 0), ext_id=-1, dt_id=-1, intr_id=-1):
        s = cls()
        vals = dict(intr=intr, R_ItoO=R_ItoO, p_IinO=p_IinO, R0=R0, p0=p0, R1=R1, p1=p1, R0_fej=R0 if R0_fej is None else R0_fej,
                    p0_fej=p0 if p0_fej is None else p0_fej, R1_fej=R1 if R1_fej is None else R1_fej,
                    p1_fej=p1 if p1_fej is None else p1_fej, w0=w0, v0=v0, w1=w1, v1=v1)
        for name, val in vals.items():
            arr = getattr(s, name)
            for i, x in enumerate(np.asarray(val, dtype=np.float64).ravel()):
                arr[i] = float(x)
        s.pose0_id, s.pose1_id, s.ext_id, s.dt_id, s.intr_id = pose0_id, pose1_id, ext_id, dt_id, intr_id
        return s


class PlvCloneSchedule(C.Structure):
    _fields_ = [("n_clones", C.c_int), ("state_time", C.c_double), ("meas_t", C.c_double), ("newest_clone_time", C.c_double),
                ("second_newest_clone_time", C.c_double), ("newest_is_imu_pose", C.c_int), ("clone_freq", C.c_int),
                ("n_sensor_times", C.c_int), ("sensor_times", C.POINTER(C.c_double)), ("sensor_dt", C.c_double),
                ("imu_oldest_t", C.c_double), ("imu_newest_t", C.c_double), ("wheel_enabled", C.c_int)]


class PlvStats(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("min", "max", "median", "mean", "rmse", "std", "ninetynine")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


ALIGN = {"posyaw": 0, "posyawsingle": 1, "se3": 2, "se3single": 3, "sim3": 4, "none": 5}


class PlvUpdateOptions(C.Structure):
    _fields_ = [("max_msckf", C.c_int), ("max_obs", C.c_int), ("chi2_mult", C.c_double), ("tri", PlvTriOptions),
                ("t_prev_frame", C.c_double), ("state_time", C.c_double), ("window_full", C.c_int),
                ("max_slam", C.c_int), ("n_slam", C.c_int), ("slam_ids", C.POINTER(C.c_uint64)), ("init_min_meas", C.c_int),
                ("cpi", C.POINTER(PlvCpiTable))]


class PlvUpdateResult(C.Structure):
    _fields_ = [("n_pool", C.c_int), ("n_msckf", C.c_int), ("n_accepted", C.c_int), ("n_rows", C.c_int),
                ("n_returned", C.c_int), ("status", C.c_int), ("n_slam", C.c_int), ("n_init", C.c_int), ("n_truncated", C.c_int)]


def select_imu_readings(t, wm, am, time0, time1):
    t, wm, am = _c64(t), _c64(wm), _c64(am)
    cap = len(t) + 2
    ot, ow, oa = np.zeros(cap), np.zeros((cap, 3)), np.zeros((cap, 3))
    n, ok = C.c_int(), C.c_int()
    rc = load_library().plv_select_imu_readings(len(t), _dp(t), _dp(wm), _dp(am), float(time0), float(time1), cap, _dp(ot), _dp(ow), _dp(oa),
                                                C.byref(n), C.byref(ok))
    assert rc == 0, rc
    return bool(ok.value), ot[:n.value].copy(), ow[:n.value].copy(), oa[:n.value].copy()


def select_wheel_data(t, m1, m2, time0, time1):
    t, m1, m2 = _c64(t), _c64(m1), _c64(m2)
    cap = len(t) + 4
    ot, o1, o2 = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    n, ok = C.c_int(), C.c_int()
    rc = load_library().plv_select_wheel_data(len(t), _dp(t), _dp(m1), _dp(m2), float(time0), float(time1), cap, _dp(ot), _dp(o1), _dp(o2),
                                              C.byref(n), C.byref(ok))
    assert rc == 0, rc
    m = n.value if ok.value else 0
    return bool(ok.value), ot[:m].copy(), o1[:m].copy(), o2[:m].copy()


def next_clone_time(n_clones, state_time, meas_t, newest, second_newest, newest_is_imu, freq, sensor_times, sensor_dt, imu_oldest,
                    imu_newest, wheel_enabled=False):
    stt = _c64(sensor_times)
    s = PlvCloneSchedule(n_clones, state_time, meas_t, newest, second_newest, 1 if newest_is_imu else 0, freq, len(stt),
                         _dp(stt) if len(stt) else None, sensor_dt, imu_oldest, imu_newest, 1 if wheel_enabled else 0)
    ct, ok = C.c_double(), C.c_int()
    rc = load_library().plv_next_clone_time(C.byref(s), C.byref(ct), C.byref(ok))
    assert rc == 0, rc
    return (ct.value if ok.value else None)


def closest_clone_time(st, t_given, exclude_newest=False):
    ct, found = C.c_double(), C.c_int()
    rc = load_library().plv_closest_clone_time(C.byref(st.c), 1 if exclude_newest else 0, float(t_given), C.byref(ct), C.byref(found))
    assert rc == 0, rc
    return (ct.value if found.value else None)


def reset_cpi(imu, clone_t):
    acc = PlvCpiAccum()
    load_library().plv_reset_cpi(C.byref(acc), C.byref(imu), float(clone_t))
    return acc


WHEEL_TYPES = {"Wheel3DAng": 0, "Wheel3DLin": 1, "Wheel3DCen": 2, "Wheel2DAng": 3, "Wheel2DLin": 4, "Wheel2DCen": 5}


class PlvIwInitOptions(C.Structure):
    _fields_ = [("wheel_type", C.c_int), ("intrinsics", C.c_double * 3), ("R_ItoO", C.c_double * 9), ("p_IinO", C.c_double * 3),
                ("toff", C.c_double), ("threshold", C.c_double), ("gravity", C.c_double * 3), ("imu_gravity_aligned", C.c_int)]


class PlvIwInitState(C.Structure):
    _fields_ = [("cnt_smooth", C.c_int), ("reserved", C.c_int), ("prev_init", C.c_double * 12)]


def init_imu_static(t, wm, am, window_time, imu_thresh, gravity=(0.0, 0.0, 9.81)):
    """I_Initializer::initialization: the 17-vector [t q p v bg ba] or None."""
    t, wm, am, g = _c64(t), _c64(wm), _c64(am), _c64(gravity)
    out, ok = np.zeros(17), C.c_int()
    rc = load_library().plv_init_imu_static(len(t), _dp(t), _dp(wm), _dp(am), float(window_time), float(imu_thresh), _dp(g), _dp(out),
                                            C.byref(ok))
    if rc != PLV_OK:
        raise PlvError(rc, "plv_init_imu_static")
    return out if ok.value else None


class IwInitializer:
    """IW_Initializer: options + the memory it keeps between attempts (plv_iw_init_state)."""

    def __init__(self, wheel_type, intrinsics, R_ItoO, p_IinO, toff, threshold, gravity=(0.0, 0.0, 9.81), imu_gravity_aligned=False):
        o = PlvIwInitOptions()
        o.wheel_type = WHEEL_TYPES[wheel_type] if isinstance(wheel_type, str) else int(wheel_type)
        for name, val in dict(intrinsics=intrinsics, R_ItoO=R_ItoO, p_IinO=p_IinO, gravity=gravity).items():
            arr = getattr(o, name)
            for i, x in enumerate(np.asarray(val, dtype=np.float64).ravel()):
                arr[i] = float(x)
        o.toff, o.threshold, o.imu_gravity_aligned = float(toff), float(threshold), 1 if imu_gravity_aligned else 0
        self.opt, self.state = o, PlvIwInitState()
        load_library().plv_iw_init_reset(C.byref(self.state))
        self.last_init, self.last_mode = None, -1

    def initialization(self, t, wm, am, tw, m1, m2):
        t, wm, am, tw, m1, m2 = _c64(t), _c64(wm), _c64(am), _c64(tw), _c64(m1), _c64(m2)
        out, init = np.zeros(17), np.full(12, np.nan)
        ok, mode = C.c_int(), C.c_int()
        rc = load_library().plv_init_imu_wheel(C.byref(self.opt), C.byref(self.state), len(t), _dp(t), _dp(wm), _dp(am), len(tw), _dp(tw),
                                               _dp(m1), _dp(m2), _dp(out), C.byref(ok), C.byref(mode), _dp(init))
        if rc != PLV_OK:
            raise PlvError(rc, "plv_init_imu_wheel")
        self.last_init, self.last_mode = (None if np.isnan(init[0]) else init), mode.value
        return out if ok.value else None


def traj_header():
    buf = C.create_string_buffer(256)
    n = load_library().plv_traj_header(buf, 256)
    assert n > 0
    return buf.value.decode()


def traj_format(t, p, q, P=None):
    buf = C.create_string_buffer(512)
    p, q = _f64(p), _f64(q)
    Pm = _f64(np.asarray(P).reshape(36)) if P is not None else None
    n = load_library().plv_traj_format(buf, 512, float(t), _dp(p), _dp(q), _dp(Pm))
    assert n > 0, n
    return buf.value.decode()


def traj_load(path):
    lib = load_library()
    n, nc = C.c_int(), C.c_int()
    rc = lib.plv_traj_load(str(path).encode(), 0, None, None, None, None, C.byref(n), C.byref(nc))
    if rc != 0:
        raise PlvError(rc, lib.plv_last_error().decode())
    N = n.value
    t, poses, co, cp = np.zeros(N), np.zeros((N, 7)), np.zeros((N, 3, 3)), np.zeros((N, 3, 3))
    rc = lib.plv_traj_load(str(path).encode(), N, _dp(t), _dp(poses), _dp(co), _dp(cp), C.byref(n), C.byref(nc))
    if rc != 0:
        raise PlvError(rc, lib.plv_last_error().decode())
    return t, poses, co[:nc.value], cp[:nc.value]


def traj_length(poses):
    poses = _c64(poses)
    return load_library().plv_traj_length(len(poses), _dp(poses))


def traj_associate(est_times, gt_times, offset=0.0, max_difference=0.02):
    et, gt = _f64(est_times), _f64(gt_times)
    ei, gi = np.zeros(max(len(et), 1), dtype=np.int32), np.zeros(max(len(et), 1), dtype=np.int32)
    m = C.c_int()
    rc = load_library().plv_traj_associate(float(offset), float(max_difference), len(et), _dp(et), len(gt), _dp(gt), _ip(ei), _ip(gi), C.byref(m))
    assert rc == 0, rc
    return ei[:m.value].copy(), gi[:m.value].copy()


class StateView:
    """Owns the numpy arrays behind a plv_state_view."""

    def __init__(self, clone_time, clone_R, clone_p, clone_state_id, R_ItoC, p_IinC, intrinsics, clone_R_fej=None,
                 clone_p_fej=None, cam_dt=0.0, extrinsic_state_id=-1, intrinsic_state_id=-1, dt_state_id=-1, sigma_pix=1.5,
                 use_pol_cov=0, intr_ori_cov=0.0, intr_pos_cov=0.0, feat_rep=0, dt_exp=0.01, use_imu_cov=0, intr_err_mlt=1.0):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.t, self.R, self.p = f(clone_time), f(clone_R).reshape(-1, 9), f(clone_p).reshape(-1, 3)
        self.Rf = f(clone_R_fej).reshape(-1, 9) if clone_R_fej is not None else self.R.copy()
        self.pf = f(clone_p_fej).reshape(-1, 3) if clone_p_fej is not None else self.p.copy()
        self.ids = np.ascontiguousarray(clone_state_id, dtype=np.int32)
        v = PlvStateView()
        v.n_clones = len(self.t)
        v.clone_time, v.clone_R, v.clone_p = _dp(self.t), _dp(self.R), _dp(self.p)
        v.clone_R_fej, v.clone_p_fej, v.clone_state_id = _dp(self.Rf), _dp(self.pf), _ip(self.ids)
        v.R_ItoC = (C.c_double * 9)(*np.asarray(R_ItoC, dtype=np.float64).ravel())
        v.p_IinC = (C.c_double * 3)(*np.asarray(p_IinC, dtype=np.float64).ravel())
        v.intrinsics = (C.c_double * 8)(*np.asarray(intrinsics, dtype=np.float64).ravel())
        v.cam_dt, v.extrinsic_state_id, v.intrinsic_state_id, v.dt_state_id = cam_dt, extrinsic_state_id, intrinsic_state_id, dt_state_id
        v.intr_order, v.dt_exp, v.sigma_pix = 3, dt_exp, sigma_pix
        v.use_pol_cov, v.intr_ori_cov, v.intr_pos_cov, v.feat_rep = use_pol_cov, intr_ori_cov, intr_pos_cov, feat_rep
        v.use_imu_cov, v.intr_err_mlt = int(use_imu_cov), float(intr_err_mlt)
        self.c = v


class Tracks:
    """Owns the numpy arrays behind a plv_tracks (CSR observation lists)."""

    def __init__(self, obs_ptr, obs_time, obs_uv, p_FinG, p_FinG_fej=None, res_R=None, res_p=None, obs_uvn=None, res_Q=None, res_clone=None):
        self.ptr = np.ascontiguousarray(obs_ptr, dtype=np.int32)
        self.t = np.ascontiguousarray(obs_time, dtype=np.float64)
        self.uv = np.ascontiguousarray(obs_uv, dtype=np.float32).reshape(-1, 2)
        self.pf = np.ascontiguousarray(p_FinG, dtype=np.float64).reshape(-1, 3)
        self.pff = np.ascontiguousarray(p_FinG_fej, dtype=np.float64).reshape(-1, 3) if p_FinG_fej is not None else self.pf.copy()
        self.rR = np.ascontiguousarray(res_R, dtype=np.float64).reshape(-1, 9) if res_R is not None else None
        self.rp = np.ascontiguousarray(res_p, dtype=np.float64).reshape(-1, 3) if res_p is not None else None
        v = PlvTracks()
        v.n_feat = len(self.ptr) - 1
        v.obs_ptr, v.obs_time, v.obs_uv = _ip(self.ptr), _dp(self.t), _fp(self.uv)
        v.p_FinG, v.p_FinG_fej, v.res_R, v.res_p = _dp(self.pf), _dp(self.pff), _dp(self.rR), _dp(self.rp)
        self.uvn = np.ascontiguousarray(obs_uvn, dtype=np.float32).reshape(-1, 2) if obs_uvn is not None else None
        v.obs_uvn = _fp(self.uvn)
        self.rQ = np.ascontiguousarray(res_Q, dtype=np.float64).reshape(-1, 36) if res_Q is not None else None
        self.rc = np.ascontiguousarray(res_clone, dtype=np.int32) if res_clone is not None else None
        v.res_Q, v.res_clone = _dp(self.rQ), _ip(self.rc)
        self.c = v


class LineTracks:
    """Owns the numpy arrays behind a plv_line_tracks."""

    def __init__(self, obs_ptr, obs_time, seg_uv, seg_uvn=None, line_FinG=None, D=None, anchor_pt=None, has_pt=None,
                 res_R=None, res_p=None, res_Q=None, res_clone=None):
        f64 = lambda a, w: np.ascontiguousarray(a, dtype=np.float64).reshape(-1, w) if a is not None else None
        f32 = lambda a, w: np.ascontiguousarray(a, dtype=np.float32).reshape(-1, w) if a is not None else None
        self.ptr = np.ascontiguousarray(obs_ptr, dtype=np.int32)
        self.t = np.ascontiguousarray(obs_time, dtype=np.float64)
        self.uv, self.uvn = f32(seg_uv, 4), f32(seg_uvn, 4)
        self.lg, self.ap = f64(line_FinG, 6), f64(anchor_pt, 3)
        self.D = np.ascontiguousarray(D, dtype=np.int32) if D is not None else None
        self.hp = np.ascontiguousarray(has_pt, dtype=np.uint8) if has_pt is not None else None
        self.rR, self.rp = f64(res_R, 9), f64(res_p, 3)
        v = PlvLineTracks()
        v.n_lines = len(self.ptr) - 1
        v.obs_ptr, v.obs_time, v.seg_uv, v.seg_uvn = _ip(self.ptr), _dp(self.t), _fp(self.uv), _fp(self.uvn)
        v.line_FinG, v.D, v.anchor_pt, v.has_pt = _dp(self.lg), _ip(self.D), _dp(self.ap), _u8p(self.hp)
        v.res_R, v.res_p = _dp(self.rR), _dp(self.rp)
        self.rQ = np.ascontiguousarray(res_Q, dtype=np.float64).reshape(-1, 36) if res_Q is not None else None
        self.rc = np.ascontiguousarray(res_clone, dtype=np.int32) if res_clone is not None else None
        v.res_Q, v.res_clone = _dp(self.rQ), _ip(self.rc)
        self.c = v


def cpi_noise(st, cpi, t_q):
    """plv_cpi_noise: (Q [n][36], clone index [n], ok [n]) of the CPI record behind each query time."""
    t_q = np.ascontiguousarray(t_q, dtype=np.float64)
    Q, ci, ok = np.zeros((len(t_q), 36)), np.zeros(len(t_q), dtype=np.int32), np.zeros(len(t_q), dtype=np.uint8)
    rc = load_library().plv_cpi_noise(C.byref(st.c), C.byref(cpi.c), len(t_q), _dp(t_q), _dp(Q), _ip(ci), _u8p(ok))
    if rc != PLV_OK:
        raise PlvError(rc, load_library().plv_last_error().decode())
    return Q, ci, ok


def jpl_left_update(q, dth=None, R=None):
    """plv_jpl_left_update in place: q [n][4] (C-contiguous float64) <- [dth / 2, 1] (x) q; R [n][9] receives the rotation matrices."""
    lib = load_library()
    n = q.shape[0] if q.ndim == 2 else 1
    d = np.ascontiguousarray(dth, dtype=np.float64) if dth is not None else None
    lib.plv_jpl_left_update(n, _dp(q), _dp(d), _dp(R))


def line_worker_config(spin_us=-1, fit_threads=-1):
    """plv_line_worker_config: (polling budget in us, fitter threads) of the library's line threads; negative = query only"""
    a, b = C.c_int(), C.c_int()
    load_library().plv_line_worker_config(int(spin_us), int(fit_threads), C.byref(a), C.byref(b))
    return a.value, b.value


def phase_counters():
    """plv_phase_counters (measurement aid): ns inside the parts of plv_camera_frame since the library was loaded"""
    out = (C.c_ulonglong * 10)()
    load_library().plv_phase_counters(out)
    return dict(zip(("flow_wait", "points", "points_wait", "lines", "line_join", "w_wake", "w_maps", "w_extract", "w_feed_start", "w_feed"),
                    [int(x) for x in out]))


def alloc_count():
    """plv_alloc_count (measurement aid): device / pinned buffer (re)allocations since the library was loaded"""
    return int(load_library().plv_alloc_count())


def route_counts():
    """plv_route_counts (measurement aid): updates collected so far by the route they took (index = update_compression_mode()[1])"""
    out = (C.c_ulonglong * 8)()
    load_library().plv_route_counts(out)
    return [int(v) for v in out]


def speculation_counts():
    """plv_speculation_counts (measurement aid): [used, used with a pool above max_msckf, withdrawn (cut by the cap), withdrawn (pool larger than the launch)]"""
    out = (C.c_ulonglong * 4)()
    load_library().plv_speculation_counts(out)
    return [int(v) for v in out]


def memory_bytes():
    """plv_memory_bytes: dict(device, pinned, device_peak, pinned_peak) — bytes the library holds, all contexts of the process"""
    out = (C.c_ulonglong * 4)()
    load_library().plv_memory_bytes(out)
    return dict(zip(("device", "pinned", "device_peak", "pinned_peak"), [int(v) for v in out]))


def memory_policy(growth_percent=-1, device_floor_kb=-1, pinned_floor_kb=-1):
    """plv_memory_policy: how generously a growing buffer is sized (process-wide); resets the peaks of memory_bytes()"""
    rc = load_library().plv_memory_policy(growth_percent, device_floor_kb, pinned_floor_kb)
    if rc != 0:
        raise PlvError(rc, "plv_memory_policy")


def device_numa_node(device=0):
    """plv_device_numa_node: NUMA node of the HIP device's PCI function, -1 when unknown"""
    return int(load_library().plv_device_numa_node(int(device)))


def chain_count():
    """plv_chain_count (measurement aid): line launches plv_camera_try_update enqueued behind a point update that was still running"""
    return int(load_library().plv_chain_count())


def debug_knobs(mask=-1):
    """plv_debug_knobs (measurement aid): sets the mask of alternative placements (csrc/plv_ctx.hpp "Measurement knobs"), returns the previous one.
    PLV_TEST_KNOBS_OR (tests): bits that stay set whatever mask a test asks for — e.g. 1 << 28, the napping helper threads, under the
    whole GPU suite (tools/suite_with_naps.sh)."""
    if mask >= 0:
        mask |= int(os.environ.get("PLV_TEST_KNOBS_OR", "0"))
    return int(load_library().plv_debug_knobs(int(mask)))


def counters():
    """plv_counters: dict(launches, syncs, copies, copy_bytes, lk_iters, lines_detected, frame_ns, sync_ns) since the library was loaded"""
    lib = load_library()
    out = (C.c_ulonglong * 8)()
    lib.plv_counters(out)
    return dict(zip(("launches", "syncs", "copies", "copy_bytes", "lk_iters", "lines_detected", "frame_ns", "sync_ns"), [int(x) for x in out]))


def default_config(width=752, height=480):
    lib = load_library()
    cfg = PlvConfig()
    lib.plv_config_default(C.byref(cfg), width, height)
    return cfg


class Context:
    """One plv_ctx (one camera / one HIP stream).  Raises PlvError on any non-OK status except
    where a method documents a returned status."""

    def __init__(self, cfg=None):
        self.lib = load_library()
        self.cfg = cfg if cfg is not None else default_config()
        h = C.c_void_p()
        rc = self.lib.plv_ctx_create(C.byref(self.cfg), C.byref(h))
        if rc != PLV_OK:
            raise PlvError(rc, self.lib.plv_last_error().decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.plv_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow=()):
        if rc != PLV_OK and rc not in allow:
            raise PlvError(rc, self.lib.plv_last_error().decode())
        return rc

    # ---- profiling
    def prof_enable(self, on=True):
        self._chk(self.lib.plv_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self._chk(self.lib.plv_prof_reset(self.h))

    def prof_table(self):
        out = {}
        for i in range(self.lib.plv_prof_count(self.h)):
            name = C.create_string_buffer(128)
            n, ms = C.c_int(), C.c_double()
            self.lib.plv_prof_get(self.h, i, name, 128, C.byref(n), C.byref(ms))
            out[name.value.decode()] = (n.value, ms.value)
        return out

    def synchronize(self):
        self._chk(self.lib.plv_ctx_synchronize(self.h))

    # ---- update side
    def cov_upload(self, P):
        P = _f64(P)
        self._chk(self.lib.plv_cov_upload(self.h, _dp(P), P.shape[0], P.shape[0]))

    def cov_download(self, n):
        P = np.zeros((n, n), order="F")
        self._chk(self.lib.plv_cov_download(self.h, _dp(P), n, n))
        return P

    def cov_checkpoint(self):
        self._chk(self.lib.plv_cov_checkpoint(self.h))

    def cov_rollback(self):
        self._chk(self.lib.plv_cov_rollback(self.h))

    def ekf_update(self, P, H, cols, res, Rdiag=None):
        """Returns (status, P_new, dx).  status is PLV_OK or PLV_E_NOT_PSD (P unchanged)."""
        P = _f64(P).copy(order="F")
        H = _f64(H)
        cols = _i32(cols)
        res = np.ascontiguousarray(res, dtype=np.float64)
        Rd = np.ascontiguousarray(Rdiag, dtype=np.float64) if Rdiag is not None else None
        n, (r, k) = P.shape[0], H.shape
        dx = np.zeros(n)
        rc = self.lib.plv_ekf_update(self.h, _dp(P), n, n, _dp(H), r, k, r, _ip(cols), _dp(res), _dp(Rd), _dp(dx))
        self._chk(rc, allow=(PLV_E_NOT_PSD,))
        return rc, P, dx

    def compress(self, H, res):
        H = _f64(H).copy(order="F")
        res = np.ascontiguousarray(res, dtype=np.float64).copy()
        m, k = H.shape
        mo = C.c_int()
        self._chk(self.lib.plv_compress(self.h, _dp(H), m, k, m, _dp(res), C.byref(mo)))
        return H[:mo.value, :].copy(), res[:mo.value].copy()

    def nullspace_batch(self, rows, Hf, Hx, res):
        """Hf [F, fdim, ld], Hx [F, k, ld], res [F, ld] C-contiguous (== col-major per feature)."""
        Hf = np.ascontiguousarray(Hf, dtype=np.float64).copy()
        Hx = np.ascontiguousarray(Hx, dtype=np.float64).copy()
        res = np.ascontiguousarray(res, dtype=np.float64).copy()
        rows = _i32(rows)
        F, fdim, ld = Hf.shape
        k = Hx.shape[1]
        self._chk(self.lib.plv_nullspace_batch(self.h, F, fdim, k, ld, _ip(rows), _dp(Hf), _dp(Hx), _dp(res)))
        return Hf, Hx, res

    def chi2_batch(self, P, rows, Hx, res, cols, sigma2):
        P = _f64(P)
        Hx = np.ascontiguousarray(Hx, dtype=np.float64)
        res = np.ascontiguousarray(res, dtype=np.float64)
        rows, cols = _i32(rows), _i32(cols)
        F, k, ld = Hx.shape
        n = P.shape[0]
        chi = np.zeros(F)
        self._chk(self.lib.plv_chi2_batch(self.h, _dp(P), n, n, F, k, ld, _ip(rows), _dp(Hx), _dp(res), _ip(cols),
                                          float(sigma2), _dp(chi)))
        return chi

    def msckf_update(self, P, rows, Hf, Hx, res, cols, sigma2, chi2_mult=1.0, res_norm_gate=3.0):
        """Returns (status, P_new, dx, accepted, n_rows)."""
        P = _f64(P).copy(order="F")
        Hf = np.ascontiguousarray(Hf, dtype=np.float64)
        Hx = np.ascontiguousarray(Hx, dtype=np.float64)
        res = np.ascontiguousarray(res, dtype=np.float64)
        rows, cols = _i32(rows), _i32(cols)
        F, fdim, ld = Hf.shape
        k = Hx.shape[1]
        n = P.shape[0]
        dx = np.zeros(n)
        acc = np.zeros(F, dtype=np.uint8)
        nrows = C.c_int()
        rc = self.lib.plv_msckf_update(self.h, _dp(P), n, n, F, fdim, k, ld, _ip(rows), _dp(Hf), _dp(Hx), _dp(res),
                                       _ip(cols), float(sigma2), float(chi2_mult), float(res_norm_gate), _u8p(acc),
                                       C.byref(nrows), _dp(dx))
        self._chk(rc, allow=(PLV_E_NOT_PSD,))
        return rc, P, dx, acc, nrows.value

    def feat_batch_upload(self, rows, Hf, Hx, res, cols):
        Hf = np.ascontiguousarray(Hf, dtype=np.float64)
        Hx = np.ascontiguousarray(Hx, dtype=np.float64)
        res = np.ascontiguousarray(res, dtype=np.float64)
        rows, cols = _i32(rows), _i32(cols)
        F, fdim, ld = Hf.shape
        k = Hx.shape[1]
        self._chk(self.lib.plv_feat_batch_upload(self.h, F, fdim, k, ld, _ip(rows), _dp(Hf), _dp(Hx), _dp(res),
                                                 _ip(cols)))
        self._batch_F = F

    def msckf_update_resident(self, n, sigma2, chi2_mult=1.0, res_norm_gate=3.0):
        dx = np.zeros(n)
        acc = np.zeros(self._batch_F, dtype=np.uint8)
        nrows = C.c_int()
        rc = self.lib.plv_msckf_update_resident(self.h, float(sigma2), float(chi2_mult), float(res_norm_gate),
                                                _u8p(acc), C.byref(nrows), _dp(dx))
        self._chk(rc, allow=(PLV_E_NOT_PSD,))
        return rc, dx, acc, nrows.value

    def update_compression_mode(self, mode=-1):
        """plv_update_compression_mode: sets (0 whitened update, 1 Householder) or queries (-1); returns (mode, route of the last
        update, ambiguous pivots its Gram factorisation met)"""
        route, amb = C.c_int(), C.c_int()
        m = self.lib.plv_update_compression_mode(self.h, int(mode), C.byref(route), C.byref(amb))
        if m < 0:
            raise PlvError(m, "plv_update_compression_mode")
        return m, route.value, amb.value

    def update_graph_mode(self, on=-1):
        c, r = C.c_int(), C.c_int()
        self._chk(self.lib.plv_update_graph_mode(self.h, int(on), C.byref(c), C.byref(r)))
        return c.value, r.value

    def msckf_update_resident_launch(self, sigma2, chi2_mult=1.0, res_norm_gate=3.0):
        self._chk(self.lib.plv_msckf_update_resident_launch(self.h, float(sigma2), float(chi2_mult), float(res_norm_gate)))

    def msckf_update_resident_wait(self, n):
        dx = np.zeros(n)
        acc = np.zeros(self._batch_F, dtype=np.uint8)
        nrows = C.c_int()
        rc = self.lib.plv_msckf_update_resident_wait(self.h, _u8p(acc), C.byref(nrows), _dp(dx))
        self._chk(rc, allow=(PLV_E_NOT_PSD,))
        return rc, dx, acc, nrows.value

    # ---- point front-end
    def feed_image(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if img.shape != (self.cfg.height, self.cfg.width):
            raise PlvError(PLV_E_BADARG, f"image shape {img.shape} != ({self.cfg.height}, {self.cfg.width})")
        self._chk(self.lib.plv_feed_image(self.h, _u8p(img), img.shape[1]))

    def image_stage(self, slot, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if img.shape != (self.cfg.height, self.cfg.width):
            raise PlvError(PLV_E_BADARG, f"image shape {img.shape} != ({self.cfg.height}, {self.cfg.width})")
        self._chk(self.lib.plv_image_stage(self.h, slot, _u8p(img), img.shape[1]))

    def feed_staged(self, slot):
        self._chk(self.lib.plv_feed_staged(self.h, slot))

    def image_buffer(self, index):
        """numpy view (height x width, uint8) of the library's page-locked image block `index` (plv_image_buffer): an image written
        into it is read by the frame's first kernel from where it lies (no host copy inside the call)"""
        ptr, stride = C.POINTER(C.c_uint8)(), C.c_int()
        self._chk(self.lib.plv_image_buffer(self.h, index, C.byref(ptr), C.byref(stride)))
        buf = (C.c_uint8 * (self.cfg.height * stride.value)).from_address(C.addressof(ptr.contents))
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.cfg.height, stride.value)[:, :self.cfg.width]

    def pyramid_levels(self, which=0):
        return self.lib.plv_pyramid_levels(self.h, which)

    def pyramid_level(self, which, level):
        w, h = C.c_int(), C.c_int()
        self._chk(self.lib.plv_pyramid_download(self.h, which, level, C.byref(w), C.byref(h), None))
        out = np.zeros((h.value, w.value), dtype=np.uint8)
        self._chk(self.lib.plv_pyramid_download(self.h, which, level, C.byref(w), C.byref(h), _u8p(out)))
        return out

    def lk_track(self, pts0, pts1_init):
        pts0 = np.ascontiguousarray(pts0, dtype=np.float32)
        pts1 = np.ascontiguousarray(pts1_init, dtype=np.float32).copy()
        n = pts0.shape[0]
        st = np.zeros(n, dtype=np.uint8)
        it = np.zeros(n, dtype=np.int32)
        self._chk(self.lib.plv_lk_track(self.h, n, _fp(pts0), _fp(pts1), _u8p(st), _ip(it)))
        return pts1, st, it

    def undistort(self, uv):
        uv = np.ascontiguousarray(uv, dtype=np.float32)
        out = np.zeros_like(uv)
        self._chk(self.lib.plv_undistort(self.h, uv.shape[0], _fp(uv), _fp(out)))
        return out

    def ransac(self, m1, m2, thr, seed=0):
        m1 = np.ascontiguousarray(m1, dtype=np.float32)
        m2 = np.ascontiguousarray(m2, dtype=np.float32)
        n = m1.shape[0]
        mask = np.zeros(n, dtype=np.uint8)
        good, it = C.c_int(), C.c_int()
        self._chk(self.lib.plv_ransac_fundamental(self.h, n, _fp(m1), _fp(m2), float(thr), seed, _u8p(mask),
                                                  C.byref(good), C.byref(it)))
        return mask, good.value, it.value

    def perform_matching_launch(self, pts0, pts1_init):
        pts0 = np.ascontiguousarray(pts0, dtype=np.float32)
        pts1 = np.ascontiguousarray(pts1_init, dtype=np.float32)
        self._match_n = pts0.shape[0]
        self._chk(self.lib.plv_perform_matching_launch(self.h, self._match_n, _fp(pts0), _fp(pts1)))

    def perform_matching_wait(self):
        n = self._match_n
        pts1, mask = np.zeros((n, 2), dtype=np.float32), np.zeros(n, dtype=np.uint8)
        n0, n1 = np.zeros((n, 2), dtype=np.float32), np.zeros((n, 2), dtype=np.float32)
        it = C.c_longlong()
        self._chk(self.lib.plv_perform_matching_wait(self.h, _fp(pts1), _u8p(mask), _fp(n0), _fp(n1), C.byref(it)))
        return pts1, mask, n0, n1, it.value

    def perform_matching(self, pts0, pts1_init):
        pts0 = np.ascontiguousarray(pts0, dtype=np.float32)
        pts1 = np.ascontiguousarray(pts1_init, dtype=np.float32).copy()
        n = pts0.shape[0]
        mask = np.zeros(n, dtype=np.uint8)
        n0 = np.zeros((n, 2), dtype=np.float32)
        n1 = np.zeros((n, 2), dtype=np.float32)
        it = C.c_longlong()
        self._chk(self.lib.plv_perform_matching(self.h, n, _fp(pts0), _fp(pts1), _u8p(mask), _fp(n0), _fp(n1),
                                                C.byref(it)))
        return pts1, mask, n0, n1, it.value

    # ---- per-feature Jacobians
    def jacobian_columns(self, st, tr, cap=1024):
        cols = np.zeros(cap, dtype=np.int32)
        k = C.c_int()
        self._chk(self.lib.plv_jacobian_columns(C.byref(st.c), C.byref(tr.c), _ip(cols), cap, C.byref(k)))
        return cols[:k.value].copy()

    def build_jacobians(self, st, tr, cols, ld):
        cols = _i32(cols)
        F, k = tr.c.n_feat, len(cols)
        rows = np.zeros(F, dtype=np.int32)
        Hf, Hx, res = np.zeros((F, 3, ld)), np.zeros((F, k, ld)), np.zeros((F, ld))
        self._chk(self.lib.plv_build_jacobians(self.h, C.byref(st.c), C.byref(tr.c), k, _ip(cols), ld, _ip(rows), _dp(Hf),
                                               _dp(Hx), _dp(res)))
        return rows, Hf, Hx, res

    def build_jacobians_resident(self, st, tr, cols, ld):
        cols = _i32(cols)
        self._chk(self.lib.plv_build_jacobians_resident(self.h, C.byref(st.c), C.byref(tr.c), len(cols), _ip(cols), ld))
        self._batch_F = tr.c.n_feat

    def propagate(self, imu, noise, t, wm, am, n, acc=None, imu_id=0, want_records=True):
        """Propagator::propagate over the given samples on the resident covariance; imu / acc are updated in place."""
        t, wm, am = _c64(t), _c64(wm), _c64(am)
        rec = (PlvCpiRecord * max(len(t) - 1, 1))() if (acc is not None and want_records) else None
        Phi, Qd = np.zeros((15, 15)), np.zeros((15, 15))
        self._chk(self.lib.plv_propagate(self.h, C.byref(imu), C.byref(noise), len(t), _dp(t), _dp(wm), _dp(am),
                                         C.byref(acc) if acc is not None else None, rec, n, imu_id, _dp(Phi), _dp(Qd)))
        return Phi, Qd, (list(rec)[:len(t) - 1] if rec is not None else [])

    def wheel_linear_system(self, opt, st, t, m1, m2):
        t, m1, m2 = _c64(t), _c64(m1), _c64(m2)
        H, res, Cov, cols, k, rows = np.zeros(22 * 6), np.zeros(6), np.zeros(36), np.zeros(22, dtype=np.int32), C.c_int(), C.c_int()
        R, p = np.zeros((3, 3)), np.zeros(3)
        self._chk(self.lib.plv_wheel_linear_system(self.h, C.byref(opt), C.byref(st), len(t), _dp(t), _dp(m1), _dp(m2), _dp(H), _dp(res),
                                                   _dp(Cov), _ip(cols), C.byref(k), C.byref(rows), _dp(R), _dp(p)))
        r, kk = rows.value, k.value
        return (H[:r * kk].reshape(kk, r).T.copy(), res[:r].copy(), Cov[:r * r].reshape(r, r).copy(), cols[:kk].copy(), R, p)   # H as rows x k

    def wheel_update(self, opt, st, t, m1, m2, n):
        t, m1, m2 = _c64(t), _c64(m1), _c64(m2)
        acc, dx = np.zeros(1, dtype=np.uint8), np.zeros(n)
        rc = self.lib.plv_wheel_update(self.h, C.byref(opt), C.byref(st), len(t), _dp(t), _dp(m1), _dp(m2), _u8p(acc), _dp(dx))
        self._chk(rc, allow=(PLV_E_NOT_PSD,))
        return rc, int(acc[0]), dx

    def cpi_integrate(self, noise, t_given, clone_t, R_clone, v_clone, bg, ba, t, wm, am):
        """State::create_new_cpi_integrate: (ok, PlvCpiRecord)."""
        t, wm, am = _c64(t), _c64(wm), _c64(am)
        Rc, vc, bg, ba = _c64(R_clone), _c64(v_clone), _c64(bg), _c64(ba)
        rec, ok = PlvCpiRecord(), C.c_int()
        self._chk(self.lib.plv_cpi_integrate(self.h, C.byref(noise), float(t_given), float(clone_t), _dp(Rc), _dp(vc), _dp(bg), _dp(ba),
                                             len(t), _dp(t), _dp(wm), _dp(am), C.byref(rec), C.byref(ok)))
        return bool(ok.value), rec

    def cov_clone(self, n, src_id, size=6):
        self._chk(self.lib.plv_cov_clone(self.h, n, src_id, size))

    def traj_ate(self, est, gt, method="posyaw", n_aligned=-1):
        est, gt = _c64(est), _c64(gt)
        n = len(est)
        R, t, s = np.zeros((3, 3)), np.zeros(3), C.c_double()
        al, oe, pe = np.zeros((n, 7)), np.zeros(n), np.zeros(n)
        so, sp = PlvStats(), PlvStats()
        self._chk(self.lib.plv_traj_ate(self.h, ALIGN[method], n, _dp(est), _dp(gt), n_aligned, _dp(R), _dp(t), C.byref(s), _dp(al),
                                        _dp(oe), _dp(pe), C.byref(so), C.byref(sp)))
        return dict(R=R, t=t, s=s.value, aligned=al, ori_err=oe, pos_err=pe, ori=so.as_dict(), pos=sp.as_dict())

    def cpi_poses(self, st, cpi, t_q):
        t_q = _f64(t_q)
        R, p, ok = np.zeros((len(t_q), 9)), np.zeros((len(t_q), 3)), np.zeros(len(t_q), dtype=np.uint8)
        self._chk(self.lib.plv_cpi_poses(self.h, C.byref(st.c), C.byref(cpi.c), len(t_q), _dp(t_q), _dp(R), _dp(p), _u8p(ok)))
        return R, p, ok

    # ---- SLAM landmarks
    def slam_update(self, n, H, res, cols, chi2_mult=1.0):
        H = np.asfortranarray(H, dtype=np.float64)      # rows x k
        rows, k = H.shape
        res, cols = _f64(res), _i32(cols)
        acc = np.zeros(1, dtype=np.uint8)
        dx = np.zeros(n)
        rc = self.lib.plv_slam_update(self.h, rows, k, rows, _dp(H), _dp(res), _ip(cols), float(chi2_mult), _u8p(acc), _dp(dx))
        self._chk(rc, allow=(PLV_E_NOT_PSD,))
        return rc, int(acc[0]), dx

    def slam_initialize(self, n, Hf, Hx, res, cols, chi2_mult=1.0):
        Hf, Hx = np.asfortranarray(Hf, dtype=np.float64), np.asfortranarray(Hx, dtype=np.float64)  # rows x 3, rows x k
        rows, k = Hx.shape
        res, cols = _f64(res), _i32(cols)
        ok = np.zeros(1, dtype=np.uint8)
        dxi, dx = np.zeros(3), np.zeros(n + 3)
        self._chk(self.lib.plv_slam_initialize(self.h, rows, k, rows, _dp(Hf), _dp(Hx), _dp(res), _ip(cols), float(chi2_mult), _u8p(ok),
                                               _dp(dxi), _dp(dx)))
        return int(ok[0]), dxi, dx

    def set_lk_window(self, win):
        """plv_set_lk_window: the LK window (plv_config.win_size) of a live context"""
        self._chk(self.lib.plv_set_lk_window(self.h, int(win)))

    def set_camera_intrinsics(self, K8):
        K8 = _c64(K8)
        self._chk(self.lib.plv_set_camera_intrinsics(self.h, _dp(K8)))

    def cov_marginalize(self, idx, size):
        self._chk(self.lib.plv_cov_marginalize(self.h, int(idx), int(size)))

    # ---- UpdaterCamera::try_update, point half
    def db_append_measurements(self, fid, t, uv, uvn):
        t = np.ascontiguousarray(t, dtype=np.float64)
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        uvn = np.ascontiguousarray(uvn, dtype=np.float32).reshape(-1, 2)
        self._chk(self.lib.plv_db_append_measurements(self.h, int(fid), len(t), _dp(t), _fp(uv), _fp(uvn)))

    def camera_update_points(self, st, n, max_msckf, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0,
                             min_dist=0.1, max_dist=60.0, max_cond=1e4, max_baseline=40.0, refine=True, max_slam=0, slam_ids=(),
                             init_min_meas=10, cpi=None):
        sl = np.ascontiguousarray(slam_ids, dtype=np.uint64)
        opt = PlvUpdateOptions(max_msckf, max_obs, chi2_mult, PlvTriOptions(min_dist, max_dist, max_cond, max_baseline, 1 if refine else 0),
                               t_prev_frame, state_time, 1 if window_full else 0, max_slam, len(sl), _u64p(sl) if len(sl) else None,
                               init_min_meas, C.pointer(cpi.c) if cpi is not None else None)
        res = PlvUpdateResult()
        dx = np.zeros(n)
        ids = np.zeros(max_msckf, dtype=np.uint64)
        acc = np.zeros(max_msckf, dtype=np.uint8)
        p = np.zeros((max_msckf, 3))
        self._chk(self.lib.plv_camera_update_points(self.h, C.byref(st.c), C.byref(opt), _dp(dx), C.byref(res), _u64p(ids), _u8p(acc),
                                                    _dp(p)))
        m = res.n_msckf
        return dict(dx=dx, n_pool=res.n_pool, n_msckf=m, n_accepted=res.n_accepted, n_rows=res.n_rows, n_returned=res.n_returned,
                    status=res.status, ids=ids[:m].copy(), accepted=acc[:m].copy(), p_FinG=p[:m].copy(), n_slam=res.n_slam,
                    n_init=res.n_init, n_truncated=res.n_truncated)

    def camera_update_list(self, which):
        """SLAM (0) / SLAM-init (1) list of the last camera_update_points: ids, obs_ptr, obs_time, obs_uv, obs_uvn, p_FinG."""
        n = C.c_int()
        self._chk(self.lib.plv_camera_update_list(self.h, which, 0, 0, C.byref(n), None, None, None, None, None, None))
        F = n.value
        ids, ptr = np.zeros(F, dtype=np.uint64), np.zeros(F + 1, dtype=np.int32)
        cap = 4096
        t, uv, uvn, p = np.zeros(cap), np.zeros((cap, 2), dtype=np.float32), np.zeros((cap, 2), dtype=np.float32), np.zeros((F, 3))
        if F:
            self._chk(self.lib.plv_camera_update_list(self.h, which, F, cap, C.byref(n), _u64p(ids), _ip(ptr), _dp(t), _fp(uv), _fp(uvn),
                                                      _dp(p)))
        m = int(ptr[-1])
        return dict(ids=ids, obs_ptr=ptr, obs_time=t[:m].copy(), obs_uv=uv[:m].copy(), obs_uvn=uvn[:m].copy(), p_FinG=p)

    def slam_marg_flags(self, slam_ids, fail_count=None):
        sl = np.ascontiguousarray(slam_ids, dtype=np.uint64)
        fc = np.ascontiguousarray(fail_count, dtype=np.int32) if fail_count is not None else None
        out = np.zeros(len(sl), dtype=np.uint8)
        self._chk(self.lib.plv_slam_marg_flags(self.h, len(sl), _u64p(sl), _ip(fc) if fc is not None else None, _u8p(out)))
        return out

    def line_db_append_measurements(self, lid, t, seg_uv, seg_uvn, D=0, point_ids=()):
        t = np.ascontiguousarray(t, dtype=np.float64)
        uv = np.ascontiguousarray(seg_uv, dtype=np.float32).reshape(-1, 4)
        uvn = np.ascontiguousarray(seg_uvn, dtype=np.float32).reshape(-1, 4)
        pid = np.ascontiguousarray(point_ids, dtype=np.int32)
        self._chk(self.lib.plv_line_db_append_measurements(self.h, int(lid), len(t), _dp(t), _fp(uv), _fp(uvn), int(D),
                                                           _ip(pid) if len(pid) else None, len(pid)))

    def point_used_insert(self, fid, p, newest):
        p = np.ascontiguousarray(p, dtype=np.float64)
        self._chk(self.lib.plv_point_used_insert(self.h, int(fid), _dp(p), float(newest)))

    def camera_get_line_features(self, st, n=None, max_obs=None, **kw):
        """plv_camera_get_line_features: records the state (before the point update's dx is applied) the next camera_update_lines
        triangulates its pool on, as the reference's try_update orders it"""
        self._chk(self.lib.plv_camera_get_line_features(self.h, C.byref(st.c)))

    def camera_update_lines(self, st, n, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0, cap=512, cpi=None):
        opt = PlvUpdateOptions(0, max_obs, chi2_mult, PlvTriOptions(0, 0, 0, 0, 0), t_prev_frame, state_time, 1 if window_full else 0,
                               0, 0, None, 10, C.pointer(cpi.c) if cpi is not None else None)
        res = PlvUpdateResult()
        dx = np.zeros(n)
        ids, acc, lg = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint8), np.zeros((cap, 6))
        self._chk(self.lib.plv_camera_update_lines(self.h, C.byref(st.c), C.byref(opt), _dp(dx), C.byref(res), _u64p(ids), _u8p(acc),
                                                   _dp(lg), cap))
        m = res.n_msckf
        return dict(dx=dx, n_pool=res.n_pool, n_lines=m, n_accepted=res.n_accepted, n_rows=res.n_rows, n_returned=res.n_returned,
                    status=res.status, ids=ids[:m].copy(), accepted=acc[:m].copy(), line_FinG=lg[:m].copy())

    def _try_update_io(self, plus, n, max_msckf, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0, min_dist=0.1, max_dist=60.0,
                       max_cond=1e4, max_baseline=40.0, refine=True, init_min_meas=10, lines=True, cap=512):
        """plv_try_update for the given options + a function that turns the filled structure into the result dicts.  The structures and
        the output arrays of a given set of options are laid out once per context; a call only writes the fields that change from frame
        to frame (the two times, the window flag, the variable list).  The results are copies."""
        key = (n, max_msckf, max_obs, chi2_mult, min_dist, max_dist, max_cond, max_baseline, bool(refine), init_min_meas, bool(lines), cap)
        cache = self.__dict__.setdefault("_io_sets", {}) if self is not None else None
        hit = cache.get(key) if cache is not None else None
        if hit is None:
            tri = PlvTriOptions(min_dist, max_dist, max_cond, max_baseline, 1 if refine else 0)
            op = PlvUpdateOptions(max_msckf, max_obs, chi2_mult, tri, t_prev_frame, state_time, 1, 0, 0, None, init_min_meas, None)
            ol = PlvUpdateOptions(0, max_obs, chi2_mult, PlvTriOptions(0, 0, 0, 0, 0), t_prev_frame, state_time, 1, 0, 0, None, 10, None)
            rp, rl = PlvUpdateResult(), PlvUpdateResult()
            bufs = (np.zeros(n), np.zeros(n), np.zeros(max_msckf, dtype=np.uint64), np.zeros(max_msckf, dtype=np.uint8), np.zeros((max_msckf, 3)),
                    np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint8), np.zeros((cap, 6)))
            a = lambda x: x.ctypes.data
            dxp, dxl, ids, acc, p, lids, lacc, lg = bufs
            io = PlvTryUpdate(C.addressof(op), C.addressof(ol) if lines else None, 0, None, a(dxp), a(dxl), C.addressof(rp), C.addressof(rl),
                              a(ids), a(acc), a(p), a(lids), a(lacc), a(lg), cap, 0)

            def results():
                m = rp.n_msckf
                pts = dict(dx=dxp.copy(), n_pool=rp.n_pool, n_msckf=m, n_accepted=rp.n_accepted, n_rows=rp.n_rows, n_returned=rp.n_returned,
                           status=rp.status, ids=ids[:m].copy(), accepted=acc[:m].copy(), p_FinG=p[:m].copy(), n_slam=rp.n_slam, n_init=rp.n_init,
                           n_truncated=rp.n_truncated)
                if not lines:
                    return pts, None, 0
                m = rl.n_msckf
                lns = dict(dx=dxl.copy(), n_pool=rl.n_pool, n_lines=m, n_accepted=rl.n_accepted, n_rows=rl.n_rows, n_returned=rl.n_returned,
                           status=rl.status, ids=lids[:m].copy(), accepted=lacc[:m].copy(), line_FinG=lg[:m].copy())
                return pts, lns, io.line_db_size
            hit = (op, ol, io, results, bufs, rp, rl, tri, [None])      # (everything io points to lives as long as the entry)
            if cache is not None:
                cache[key] = hit
        op, ol, io, results = hit[0], hit[1], hit[2], hit[3]
        wf = 1 if window_full else 0
        op.t_prev_frame = ol.t_prev_frame = t_prev_frame
        op.state_time = ol.state_time = state_time
        op.window_full = ol.window_full = wf
        hit[8][0] = plus                     # (the variable list must outlive the call)
        results.keep = hit                   # (... and so must everything io points to, also without a context to cache it in)
        io.n_var = plus.n if plus is not None else 0
        io.vars = C.addressof(plus.vars) if plus is not None else None
        io.line_db_size = 0
        return io, results

    def camera_try_update(self, st, plus, n, max_msckf, max_obs, t_prev_frame, state_time, **kw):
        """plv_camera_try_update: the point update, its dx applied through `plus` (a BoxPlus whose arrays back `st`), then the line
        update on the updated state and its dx applied.  Returns (points dict, lines dict or None, line database size after the feed)
        in the form of camera_update_points / camera_update_lines."""
        io, results = self._try_update_io(plus, n, max_msckf, max_obs, t_prev_frame, state_time, **kw)
        self._chk(self.lib.plv_camera_try_update(self.h, C.byref(st.c), C.byref(io)))
        return results()

    def camera_frame_prepare(self, st, timestamp, slot=None, img=None, mask=None, use_lines=False, update=None):
        """The arguments of plv_camera_frame marshalled into their C structures (what a C++ caller holds already): returns the record
        camera_frame_run / camera_frame_collect take."""
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
        if slot is None:
            img = np.ascontiguousarray(img, dtype=np.uint8)
            if img.shape != (self.cfg.height, self.cfg.width):
                raise PlvError(PLV_E_BADARG, f"image shape {img.shape} != ({self.cfg.height}, {self.cfg.width})")
        io, results = self._try_update_io(**update) if update is not None else (None, None)
        f = PlvCameraFrameIo(float(timestamp), -1 if slot is None else int(slot), img.ctypes.data if slot is None else None,
                             self.cfg.width, m.ctypes.data if m is not None else None, 1 if use_lines else 0,
                             C.addressof(io) if io is not None else None, 0)
        return dict(st=st, st_ref=C.byref(st.c), f=f, f_ref=C.byref(f), io=io, results=results, keep=(img, m), rc=None, rc_sync=None)

    def camera_frame_run(self, prep, sync=False):
        """plv_camera_frame (+ plv_ctx_synchronize): the C-ABI calls and nothing else (bench.py times exactly this)"""
        prep["rc"] = self.lib.plv_camera_frame(self.h, prep["st_ref"], prep["f_ref"])
        if sync:
            prep["rc_sync"] = self.lib.plv_ctx_synchronize(self.h)

    def camera_frame_collect(self, prep):
        self._chk(prep["rc"])
        if prep["rc_sync"] is not None:
            self._chk(prep["rc_sync"])
        if prep["results"] is None:
            return None, None, prep["f"].line_db_size
        pts, lns, _ = prep["results"]()
        return pts, lns, prep["f"].line_db_size

    def camera_frame(self, st, timestamp, slot=None, img=None, mask=None, use_lines=False, update=None):
        """plv_camera_frame: tracker feed (+ vanishing points and line tracker feed) and, with update = dict(plus=, n=, max_msckf=,
        max_obs=, t_prev_frame=, state_time=, ...) (the arguments of camera_try_update), the whole of try_update.  Returns
        (points dict, lines dict, line database size) — (None, None, size) without an update."""
        prep = self.camera_frame_prepare(st, timestamp, slot, img, mask, use_lines, update)
        self.camera_frame_run(prep)
        return self.camera_frame_collect(prep)

    # ---- lines (front-end)
    def detect_lines(self, which=0, cap=4096):
        lines = np.zeros((cap, 4), dtype=np.float32)
        n = C.c_int()
        self._chk(self.lib.plv_detect_lines(self.h, which, _fp(lines), cap, C.byref(n)))
        return lines[:n.value].copy()

    def line_detect_launch(self, which=0):
        self._chk(self.lib.plv_line_detect_launch(self.h, which))

    def line_detect_finish(self, which=0):
        self._chk(self.lib.plv_line_detect_finish(self.h, which))

    def line_tracker_feed_async(self, timestamp, vps):
        v = _c64(vps)
        self._chk(self.lib.plv_line_tracker_feed_async(self.h, float(timestamp), _dp(v)))

    def line_tracker_feed_wait(self):
        self._chk(self.lib.plv_line_tracker_feed_wait(self.h))

    DECISION_VALUES = ("n_obs", "triangulated", "reproj_px", "gate_passed", "tri_cond", "tri_depth", "refined_depth", "baseline_ratio", "chi2",
                       "chi2_threshold", "res_norm")

    def decision_trace(self, on=True):
        """plv_decision_trace (test aid): keep the values behind the point update's verdicts (last_point_decisions)"""
        self._chk(self.lib.plv_decision_trace(self.h, 1 if on else 0))

    def last_point_decisions(self):
        """plv_last_point_decisions: (ids [n], values [n][11], columns = Context.DECISION_VALUES) of the last point update's pool"""
        n = C.c_int(0)
        self._chk(self.lib.plv_last_point_decisions(self.h, None, None, 0, C.byref(n)))
        ids, vals = np.zeros(n.value, dtype=np.uint64), np.zeros((n.value, len(self.DECISION_VALUES)))
        if n.value:
            self._chk(self.lib.plv_last_point_decisions(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), vals.ctypes.data_as(C.POINTER(C.c_double)),
                                                        n.value, C.byref(n)))
        return ids, vals

    def last_line_decisions(self):
        """plv_last_line_decisions: (ids [n], values [n][3] = chi2, threshold, residual norm) of the last line update's batch"""
        n = C.c_int(0)
        self._chk(self.lib.plv_last_line_decisions(self.h, None, None, 0, C.byref(n)))
        ids, vals = np.zeros(n.value, dtype=np.uint64), np.zeros((n.value, 3))
        if n.value:
            self._chk(self.lib.plv_last_line_decisions(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), vals.ctypes.data_as(C.POINTER(C.c_double)),
                                                       n.value, C.byref(n)))
        return ids, vals

    def line_prefetch_mode(self, on):
        self._chk(self.lib.plv_line_prefetch_mode(self.h, 1 if on else 0))

    def line_walk_mode(self, on_device):
        self._chk(self.lib.plv_line_walk_mode(self.h, 1 if on_device else 0))

    def assign_points_to_lines(self, lines, pts, ids):
        lines = np.ascontiguousarray(lines, dtype=np.float32).reshape(-1, 4)
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        nl, npt = len(lines), len(pts)
        kept = np.zeros(max(nl, 1), dtype=np.int32)
        rel_ptr, pos_ptr = np.zeros(nl + 1, dtype=np.int32), np.zeros(nl + 1, dtype=np.int32)
        rel_id, rel_d = np.zeros(nl * npt + 1, dtype=np.uint64), np.zeros(nl * npt + 1)
        pos = np.zeros((nl * npt + 1, 2), dtype=np.float32)
        nk = C.c_int()
        self._chk(self.lib.plv_assign_points_to_lines(_fp(lines), nl, _fp(pts), _u64p(ids), npt, _ip(kept), _ip(rel_ptr), _u64p(rel_id),
                                                      _dp(rel_d), _ip(pos_ptr), _fp(pos), C.byref(nk)))
        nk = nk.value
        return dict(kept=kept[:nk].copy(), rel_ptr=rel_ptr[:nk + 1].copy(), rel_id=rel_id[:rel_ptr[nk]].copy(),
                    rel_dist=rel_d[:rel_ptr[nk]].copy(), pos_ptr=pos_ptr[:nk + 1].copy(), pos=pos[:pos_ptr[nk]].copy())

    def line_match(self, lines_new, rel_ptr_new, rel_id_new, lines_last, rel_ptr_last, rel_id_last):
        ln = np.ascontiguousarray(lines_new, dtype=np.float32).reshape(-1, 4)
        ll = np.ascontiguousarray(lines_last, dtype=np.float32).reshape(-1, 4)
        rpn, rpl = _i32(rel_ptr_new), _i32(rel_ptr_last)
        rin = np.ascontiguousarray(rel_id_new, dtype=np.uint64)
        ril = np.ascontiguousarray(rel_id_last, dtype=np.uint64)
        out = np.zeros(max(1, len(ln)), dtype=np.int32)
        self._chk(self.lib.plv_line_match(_fp(ln), len(ln), _ip(rpn), _u64p(rin), _fp(ll), len(ll), _ip(rpl), _u64p(ril), _ip(out)))
        return out[:len(ln)].copy()

    def line_classification(self, line, vps):
        line = np.ascontiguousarray(line, dtype=np.float32)
        vps = np.ascontiguousarray(vps, dtype=np.float64)
        return self.lib.plv_line_classification(_fp(line), _dp(vps))

    def vanishing_points(self, R_ItoC, K8):
        R = np.ascontiguousarray(R_ItoC, dtype=np.float64)  # row-major
        K = np.ascontiguousarray(K8, dtype=np.float64)
        out = np.zeros((3, 2))
        self._chk(self.lib.plv_vanishing_points(_dp(R), _dp(K), _dp(out)))
        return out

    def line_tracker_feed(self, timestamp, vps):
        vps = np.ascontiguousarray(vps, dtype=np.float64)
        self._chk(self.lib.plv_line_tracker_feed(self.h, float(timestamp), _dp(vps)))

    def line_tracker_feed_points(self, timestamp, vps, pts, ids):
        vps = _c64(vps)
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        self._chk(self.lib.plv_line_tracker_feed_points(self.h, float(timestamp), _dp(vps), len(ids), _fp(pts), _u64p(ids)))

    def line_tracker_last(self, cap=4096):
        lines = np.zeros((cap, 4), dtype=np.float32)
        ids = np.zeros(cap, dtype=np.uint64)
        n = C.c_int()
        self._chk(self.lib.plv_line_tracker_last(self.h, _fp(lines), _u64p(ids), cap, C.byref(n)))
        return lines[:n.value].copy(), ids[:n.value].copy()

    def line_db_size(self):
        return self.lib.plv_line_db_size(self.h)

    def line_db_ids(self, cap=65536):
        ids = np.zeros(cap, dtype=np.uint64)
        n = C.c_int()
        self._chk(self.lib.plv_line_db_ids(self.h, _u64p(ids), cap, C.byref(n)))
        return ids[:n.value].copy()

    def line_db_export(self, ids, obs_cap=65536, pts_cap=262144):
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        n = len(ids)
        obs_ptr, pts_ptr = np.zeros(n + 1, dtype=np.int32), np.zeros(n + 1, dtype=np.int32)
        t = np.zeros(obs_cap)
        uv, uvn = np.zeros((obs_cap, 4), dtype=np.float32), np.zeros((obs_cap, 4), dtype=np.float32)
        D = np.zeros(max(n, 1), dtype=np.int32)
        pid = np.zeros(pts_cap, dtype=np.int32)
        self._chk(self.lib.plv_line_db_export_tracks(self.h, _u64p(ids), n, _ip(obs_ptr), _dp(t), _fp(uv), _fp(uvn), obs_cap, _ip(D),
                                                     _ip(pts_ptr), _ip(pid), pts_cap))
        no, npt = obs_ptr[n], pts_ptr[n]
        return dict(obs_ptr=obs_ptr, obs_time=t[:no].copy(), seg_uv=uv[:no].copy(), seg_uvn=uvn[:no].copy(), D=D[:n].copy(),
                    pts_ptr=pts_ptr, pt_ids=pid[:npt].copy())

    def line_db_remove(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        self._chk(self.lib.plv_line_db_remove(self.h, _u64p(ids), len(ids)))

    # ---- lines (update side)
    def line_jacobian_columns(self, st, lt, cap=512):
        cols = np.zeros(cap, dtype=np.int32)
        k = C.c_int()
        self._chk(self.lib.plv_line_jacobian_columns(C.byref(st.c), C.byref(lt.c), _ip(cols), cap, C.byref(k)))
        return cols[:k.value].copy()

    def build_line_jacobians(self, st, lt, cols, ld):
        cols = _i32(cols)
        L, k = lt.c.n_lines, len(cols)
        rows = np.zeros(L, dtype=np.int32)
        Hf, Hx, res = np.zeros((L, 6, ld)), np.zeros((L, k, ld)), np.zeros((L, ld))
        self._chk(self.lib.plv_build_line_jacobians(self.h, C.byref(st.c), C.byref(lt.c), k, _ip(cols), ld, _ip(rows), _dp(Hf),
                                                    _dp(Hx), _dp(res)))
        return rows, Hf, Hx, res

    def build_line_jacobians_resident(self, st, lt, cols, ld):
        cols = _i32(cols)
        self._chk(self.lib.plv_build_line_jacobians_resident(self.h, C.byref(st.c), C.byref(lt.c), len(cols), _ip(cols), ld))
        self._batch_F = lt.c.n_lines

    def triangulate_lines(self, st, lt):
        L = lt.c.n_lines
        out, ok = np.zeros((L, 6)), np.zeros(L, dtype=np.uint8)
        self._chk(self.lib.plv_triangulate_lines(self.h, C.byref(st.c), C.byref(lt.c), _dp(out), _u8p(ok)))
        return out, ok

    # ---- detection
    def perform_detection(self, which, pts, ids, currid, mask=None, cap=None):
        """Returns (pts, ids, currid) after TrackKLT::perform_detection_monocular."""
        n_in = len(pts)
        cap = cap or (n_in + 4 * self.cfg.num_features + 64)
        P = np.zeros((cap, 2), dtype=np.float32)
        I = np.zeros(cap, dtype=np.uint64)
        P[:n_in] = pts
        I[:n_in] = ids
        cid = C.c_uint64(currid)
        n_out = C.c_int()
        m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
        self._chk(self.lib.plv_perform_detection(self.h, which, _u8p(m), _fp(P), I.ctypes.data_as(C.POINTER(C.c_uint64)), n_in,
                                                 cap, C.byref(cid), C.byref(n_out)))
        return P[:n_out.value].copy(), I[:n_out.value].copy(), cid.value

    # ---- tracker frame logic + feature database
    def tracker_feed(self, timestamp, img, mask=None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
        self._chk(self.lib.plv_tracker_feed(self.h, float(timestamp), _u8p(img), img.shape[1], _u8p(m)))

    def tracker_detect_ahead(self, on):
        self._chk(self.lib.plv_tracker_detect_ahead(self.h, int(on)))

    def tracker_feed_staged(self, timestamp, slot, mask=None):
        m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
        self._chk(self.lib.plv_tracker_feed_staged(self.h, float(timestamp), int(slot), _u8p(m)))

    def tracker_feed_downsampled(self, timestamp, img, mask=None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
        self._chk(self.lib.plv_tracker_feed_downsampled(self.h, float(timestamp), _u8p(img), img.shape[1], img.shape[1], img.shape[0],
                                                        _u8p(m), img.shape[1]))

    def downsample(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        out = np.zeros((h // 2, w // 2), dtype=np.uint8)
        self._chk(self.lib.plv_downsample(self.h, _u8p(img), w, w, h, _u8p(out), w // 2))
        return out

    def feed_image_downsampled(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        self._chk(self.lib.plv_feed_image_downsampled(self.h, _u8p(img), img.shape[1], img.shape[1], img.shape[0]))

    def tracker_last(self, cap=8192):
        pts = np.zeros((cap, 2), dtype=np.float32)
        ids = np.zeros(cap, dtype=np.uint64)
        n = C.c_int()
        self._chk(self.lib.plv_tracker_last(self.h, _fp(pts), ids.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(n)))
        return pts[:n.value].copy(), ids[:n.value].copy()

    def db_size(self):
        return self.lib.plv_db_size(self.h)

    def db_select(self, mode, t, cap=65536):
        ids = np.zeros(cap, dtype=np.uint64)
        n = C.c_int()
        self._chk(self.lib.plv_db_select(self.h, mode, float(t), ids.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(n)))
        return ids[:n.value].copy()

    def db_export(self, ids, cap_obs=1 << 20):
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        ptr = np.zeros(len(ids) + 1, dtype=np.int32)
        t = np.zeros(cap_obs)
        uv = np.zeros((cap_obs, 2), dtype=np.float32)
        uvn = np.zeros((cap_obs, 2), dtype=np.float32)
        self._chk(self.lib.plv_db_export_tracks(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), len(ids), _ip(ptr), _dp(t),
                                                _fp(uv), _fp(uvn), cap_obs))
        n = ptr[-1]
        return ptr, t[:n].copy(), uv[:n].copy(), uvn[:n].copy()

    def db_cleanup_measurements(self, t):
        self._chk(self.lib.plv_db_cleanup_measurements(self.h, float(t)))

    def db_remove(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        self._chk(self.lib.plv_db_remove(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), len(ids)))

    def triangulate(self, st, tr, min_dist=0.1, max_dist=60.0, max_cond=1e4, max_baseline=40.0, refine=True):
        opt = PlvTriOptions(min_dist, max_dist, max_cond, max_baseline, 1 if refine else 0)
        F = tr.c.n_feat
        p = np.zeros((F, 3))
        ok = np.zeros(F, dtype=np.uint8)
        err = np.zeros(F)
        self._chk(self.lib.plv_triangulate(self.h, C.byref(st.c), C.byref(tr.c), C.byref(opt), _dp(p), _u8p(ok), _dp(err)))
        return p, ok, err
