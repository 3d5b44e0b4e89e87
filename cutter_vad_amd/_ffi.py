"""ctypes binding of ``include/vad_engine.h`` (the C ABI of the HIP engine).

Loading fails loudly when ``libvad_engine.so`` is missing — there is no Python or CPU
substitute for it.  The library itself loads without a GPU (so the CPU test-suite can check
its exports and the host-side weight packer); creating an engine without a usable gfx950
device fails with ``VAD_ERR_NO_DEVICE``.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvad_engine.so")

VAD_OK = 0
VAD_ERR_INVALID_ARG = -1
VAD_ERR_NO_DEVICE = -2
VAD_ERR_BAD_WEIGHTS = -3
VAD_ERR_HIP = -4
VAD_ERR_NO_SLOT = -5
VAD_ERR_BAD_SLOT = -6
VAD_ERR_UNSUPPORTED = -7
VAD_ERR_BUSY = -8

VAD_FMT_F32, VAD_FMT_I16_32767, VAD_FMT_I16_32768 = 0, 1, 2
VAD_EV_START, VAD_EV_END, VAD_EV_CONTINUE = 1, 2, 4
VAD_FRAME_SAMPLES = 512
VAD_STATE_FLOATS = 256
VAD_STREAM_SAVE_BYTES = 1120


class EngineDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("model_version", C.c_int32),
        ("weights", C.c_void_p),
        ("weights_len", C.c_size_t),
        ("device_id", C.c_int32),
        ("max_streams", C.c_int32),
        ("sample_rate", C.c_int32),
        ("flags", C.c_uint32),
    ]


class EngineInfo(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("abi_version", C.c_int32),
        ("model_version", C.c_int32),
        ("device_id", C.c_int32),
        ("max_streams", C.c_int32),
        ("open_streams", C.c_int32),
        ("compute_units", C.c_int32),
        ("streams_per_workgroup", C.c_int32),
        ("weight_bytes_device", C.c_int64),
        ("state_bytes_device", C.c_int64),
        ("steps", C.c_int64),
        ("frames", C.c_int64),
        ("device_name", C.c_char * 64),
        ("arch", C.c_char * 32),
        ("frame_samples", C.c_int32),
        ("sample_rate", C.c_int32),
    ]


class Thresholds(C.Structure):
    _fields_ = [
        ("start_probability", C.c_double),
        ("end_probability", C.c_double),
        ("start_ratio", C.c_double),
        ("end_ratio", C.c_double),
        ("start_frame_count", C.c_int32),
        ("end_frame_count", C.c_int32),
    ]


class TickResult(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n", C.c_int64),
        ("slots", C.POINTER(C.c_int64)),
        ("probs", C.POINTER(C.c_float)),
        ("events", C.POINTER(C.c_uint8)),
        ("seg_frames", C.POINTER(C.c_int32)),
        ("group_start", C.c_int64 * 13),
        ("group_frames", C.c_void_p * 12),
        ("nsamples", C.POINTER(C.c_int32)),
        ("host_us", C.c_float * 3),
        ("dropped", C.c_int64),
        ("staged_next", C.c_int64),
    ]


class TickWork(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_slots", C.c_int64),
        ("last_prob", C.POINTER(C.c_float)),
        ("frames_done", C.POINTER(C.c_int64)),
        ("active", C.POINTER(C.c_uint8)),
        ("continue_cb", C.POINTER(C.c_uint8)),
        ("continue_payload", C.POINTER(C.c_uint8)),
        ("n_work", C.c_int64),
        ("work_index", C.POINTER(C.c_int32)),
        ("work_kind", C.POINTER(C.c_uint8)),
        ("work_samples", C.POINTER(C.c_int64)),
    ]


VAD_WORK_START, VAD_WORK_END, VAD_WORK_CONTINUE, VAD_WORK_PAYLOAD, VAD_WORK_LONG = 1, 2, 4, 8, 16

# name -> (restype, argtypes); mirrors include/vad_engine.h one-to-one
_vp, _i64p, _f32p, _u8p, _i32p = C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
SIGNATURES = {
    "vad_engine_create": (C.c_int, [C.POINTER(EngineDesc), C.POINTER(_vp)]),
    "vad_engine_destroy": (None, [_vp]),
    "vad_last_error": (C.c_char_p, [_vp]),
    "vad_last_create_error": (C.c_char_p, []),
    "vad_engine_info": (C.c_int, [_vp, C.POINTER(EngineInfo)]),
    "vad_stream_open": (C.c_int, [_vp, _i64p]),
    "vad_stream_open_many": (C.c_int, [_vp, C.c_int64, _i64p]),
    "vad_stream_close": (C.c_int, [_vp, C.c_int64]),
    "vad_stream_reset": (C.c_int, [_vp, _i64p, C.c_int64]),
    "vad_stream_get_state": (C.c_int, [_vp, C.c_int64, _f32p]),
    "vad_stream_set_state": (C.c_int, [_vp, C.c_int64, _f32p]),
    "vad_stream_set_thresholds": (C.c_int, [_vp, C.c_int64, C.POINTER(Thresholds)]),
    "vad_stream_set_thresholds_many": (C.c_int, [_vp, _i64p, C.c_int64, C.POINTER(Thresholds), C.c_int64]),
    "vad_stream_save": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64]),
    "vad_stream_restore": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64]),
    "vad_step": (C.c_int, [_vp, _i64p, C.c_int64, _vp, C.c_int, C.c_float, _f32p]),
    "vad_step_events": (C.c_int, [_vp, _i64p, C.c_int64, _vp, C.c_int, C.c_float, _f32p, _u8p, _i32p]),
    "vad_step_multi": (C.c_int, [_vp, _i64p, C.c_int64, C.c_int32, _vp, C.c_int, C.c_float, _f32p, _u8p]),
    "vad_step_device": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int, C.c_float, _vp, _vp, _vp, _vp]),
    "vad_step_multi_device": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, _vp, C.c_int, C.c_float, _vp, _vp, _vp, _vp]),
    "vad_step_submit": (C.c_int, [_vp, _i64p, C.c_int64, C.c_int32, _vp, C.c_int, C.c_float, _i64p]),
    "vad_step_collect": (C.c_int, [_vp, C.c_int64, _f32p, _u8p, _i32p]),
    "vad_tick_push": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, C.c_int, C.c_int]),
    "vad_tick_push_rate": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, C.c_int, C.c_int, C.c_int32]),
    "vad_tick_push_rate_gather": (C.c_int, [_vp, _i64p, C.c_int64, C.POINTER(C.c_char_p), C.c_int32, C.c_int, C.c_int, C.c_int32, _i32p]),
    "vad_tick_cancel": (C.c_int, [_vp, C.c_int64]),
    "vad_tick_push_many": (C.c_int, [_vp, _i64p, C.c_int64, _vp, C.c_int32, C.c_int, C.c_int]),
    "vad_tick_enable_segments": (C.c_int, [_vp, C.c_int]),
    "vad_tick_take_segment": (C.c_int, [_vp, C.c_int64, _f32p, C.c_int64, _i64p]),
    "vad_tick_run": (C.c_int, [_vp, C.c_float, C.POINTER(TickResult)]),
    "vad_tick_run_work": (C.c_int, [_vp, C.c_float, C.POINTER(TickResult), C.POINTER(TickWork)]),
    "vad_tick_take_segment_wav16": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int64, _i64p]),
    "vad_step_rates_device": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_void_p), _i64p, C.POINTER(C.c_int32), _vp, C.c_float,
                                        _vp, _vp, _vp, _vp]),
    "vad_step_rates": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_void_p), _i64p, C.POINTER(C.c_int32), _i64p, C.c_float,
                                 _f32p, _u8p, _i32p]),
    "vad_resample": (C.c_int, [_vp, _f32p, C.c_int64, C.c_int32, C.c_int32, _f32p]),
    "vad_resample_multi_device": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_void_p), _i64p, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.POINTER(C.c_void_p), _vp]),
    "vad_resample_device": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, C.c_int32, _vp, _vp]),
    "vad_debug_pack_weights": (C.c_int, [C.c_int32, _vp, C.c_size_t, _f32p, C.c_size_t, C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_uint32)]),
    "vad_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vad_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vad_tick_pending": (C.c_int, [_vp, C.c_int64, _i64p]),
    "vad_tick_push_gather": (C.c_int, [_vp, _i64p, C.c_int64, C.POINTER(C.c_char_p), C.c_int32, C.c_int, C.c_int, _i32p]),
    "vad_tick_push_status": (C.c_int, [_vp, _i64p, C.c_int64, _vp, C.c_int32, C.c_int, C.c_int, _i32p]),
    "vad_tick_segment_save": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _i64p]),
    "vad_tick_segment_restore": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64]),
    "vad_resample_generic": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, _f32p]),
    "vad_resample_generic_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, _vp]),
    "vad_debug_resample_operator": (C.c_int, [C.c_int32, _f32p, C.c_size_t]),
    "vad_debug_resample_path": (C.c_int, [_vp, C.c_int]),
    "vad_debug_resample_generic_entries": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_size_t]),
    "vad_debug_pack_resample": (C.c_int, [C.c_int32, _f32p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32)]),
    "vad_debug_pack_resample_t16": (C.c_int, [C.c_int32, _f32p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32),
                                              C.POINTER(C.c_uint32)]),
    "vad_debug_sm_replay": (C.c_int, [_vp, C.c_int64, _f32p, C.c_int64, _u8p, _i32p]),
    "vad_engine_synchronize": (C.c_int, [_vp]),
    "vad_debug_set_tile": (C.c_int, [_vp, C.c_int32]),
}

_lib = None


def lib() -> C.CDLL:
    """Load the in-tree HIP engine library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m cutter_vad_amd._build` "
                "(hipcc, gfx950). The engine has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib
