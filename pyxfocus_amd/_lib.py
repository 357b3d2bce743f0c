"""ctypes binding of libpxf_hip.so (include/pxf.h) -- the only way the Python layer computes.

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises.  The
library is built in-tree by ``__graft_entry__.build()`` / ``make -C pyxfocus_amd/csrc``.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- must load first: libpxf_hip.so binds to the HIP runtime torch loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# PXF_LIB: development override (A/B of build variants); the product path is the in-tree library
LIB_PATH = os.environ.get("PXF_LIB") or os.path.join(_HERE, "csrc", "libpxf_hip.so")

PXF_NMOM = 12
SRC_KINDS = {"pointsource": 0, "circularbeam": 1, "annulus": 2, "subannulus": 3, "rectbeam": 4, "xslit": 5,
             "rectArray": 6, "convergingbeam": 7, "convergingbeam2": 8, "gaussianBeam": 9, "fanBeam": 10,
             "circFan": 11}

_P = C.c_void_p          # device pointer (tensor.data_ptr()) or NULL
_d, _i64, _i32, _u64 = C.c_double, C.c_int64, C.c_int32, C.c_uint64
_DP = C.POINTER(C.c_double)   # HOST double array
_IP = C.POINTER(C.c_int32)    # HOST int32 array
_PP = C.POINTER(C.c_void_p)   # HOST array of device pointers

# name -> argtypes (restype is int unless listed in _RESTYPE)
_SIGS = {
    "pxf_abi_version": [],
    "pxf_strerror": [C.c_int],
    "pxf_last_hip_error": [],
    "pxf_device_check": [],
    "pxf_selftest_exact_division": [_i64, _u64, _P, _P],
    "pxf_flat": [_P] * 9 + [_i64, _P, _P],
    "pxf_flatopd": [_P] * 10 + [_i64, _d, _P, _P],
    "pxf_conic": [_P] * 9 + [_i64, _d, _d, _P, _P],
    "pxf_conicopd": [_P] * 10 + [_i64, _d, _d, _d, _P, _P],
    "pxf_tracesphere": [_P] * 9 + [_i64, _d, _P, _P],
    "pxf_tracesphereopd": [_P] * 10 + [_i64, _d, _d, _P, _P],
    "pxf_tracecyl": [_P] * 9 + [_i64, _d, _P, _P],
    "pxf_tracecylopd": [_P] * 10 + [_i64, _d, _d, _P, _P],
    "pxf_cylconic": [_P] * 9 + [_i64, _d, _d, _P, _P],
    "pxf_paraxial": [_P] * 4 + [_i64, _d, _P, _P],
    "pxf_paraxialy": [_P] * 2 + [_i64, _d, _P, _P],
    "pxf_woltersine": [_P] * 9 + [_i64, _d, _d, _d, _d, _P, _P],
    "pxf_wolterprimll": [_P] * 9 + [_i64] + [_d] * 5 + [_DP, _IP, _IP, _i32, _P, _P],
    "pxf_woltersecll": [_P] * 9 + [_i64] + [_d] * 6 + [_DP, _IP, _IP, _i32, _P, _P],
    "pxf_ellipsoidwoltll": [_P] * 9 + [_i64] + [_d] * 7 + [_DP, _IP, _IP, _i32, _P, _P],
    "pxf_wsprimary": [_P] * 9 + [_i64, _d, _d, _d, _P, _P],
    "pxf_wssecondary": [_P] * 9 + [_i64, _d, _d, _d, _P, _P],
    "pxf_wsprimaryback": [_P] * 9 + [_i64, _d, _d, _d, _d, _P, _P],
    "pxf_wssecondaryback": [_P] * 9 + [_i64, _d, _d, _d, _d, _P, _P],
    "pxf_torus": [_P] * 9 + [_i64, _d, _d, _P, _P],
    "pxf_conicplus": [_P] * 9 + [_i64, _d, _d, _DP, _i32, _P, _P],
    "pxf_conicplusopd": [_P] * 10 + [_i64, _d, _d, _DP, _i32, _d, _P, _P],
    "pxf_legsurf": [_P] * 5 + [_i64, _d, _d, _d, _DP, _IP, _IP, _i32, _P, _P],
    "pxf_tracezernconic": [_P] * 10 + [_i64, _d, _d, _DP, _IP, _IP, _i32, _d, _d, _P, _P],
    "pxf_tracezernrot": [_P] * 9 + [_i64, _DP, _IP, _IP, _i32, _DP, _IP, _IP, _i32, _d, _d, _P, _P],
    "pxf_zernphase": [_P] * 6 + [_i64, _DP, _IP, _IP, _i32, _d, _d, _P, _P],
    "pxf_spocone": [_P] * 9 + [_i64, _d, _d, _P, _P],
    "pxf_grat": [_P] * 3 + [_i64, _d, _d, _P, _P, _P],
    "pxf_refract": [_P] * 6 + [_i64, _d, _d, _P, _P],
    "pxf_wolter_params": [_d, _d, _d, _DP, _DP, _DP],
    "pxf_wolterprimary": [_P] * 9 + [_i64, _d, _d, _d, _P, _P],
    "pxf_wolterprimaryopd": [_P] * 10 + [_i64, _d, _d, _d, _d, _P, _P],
    "pxf_woltersecondary": [_P] * 9 + [_i64, _d, _d, _d, _P, _P],
    "pxf_tracezern": [_P] * 9 + [_i64, _DP, _IP, _IP, _i32, _d, _P, _P],
    "pxf_tracezernopd": [_P] * 10 + [_i64, _DP, _IP, _IP, _i32, _d, _d, _P, _P],
    "pxf_transform": [_P] * 9 + [_i64] + [_d] * 6 + [_P, _P],
    "pxf_itransform": [_P] * 9 + [_i64] + [_d] * 6 + [_P, _P],
    "pxf_reflect": [_P] * 6 + [_i64, _P, _P],
    "pxf_radgrat": [_P] * 5 + [_d, _i64, _d, _d, _P, _P],
    "pxf_radgratw": [_P] * 6 + [_i64, _d, _d, _P, _P],
    "pxf_applyt": [_P] * 18 + [_i64, _DP, _DP, _P],
    "pxf_applytpos": [_P] * 6 + [_i64, _DP, _P],
    "pxf_pointto": [_P] * 6 + [_i64, _d, _d, _d, _d, _P],
    "pxf_compact_workspace_bytes": [_i64],
    "pxf_vignette_mask": [_P] * 3 + [_i64, _P, _P],
    "pxf_compact_scan": [_P, _i64, _P, _P, _P],
    "pxf_compact_scatter": [_PP, _PP, _i32, _i64, _P, _P, _P],
    "pxf_gather": [_PP, _PP, _i32, _P, _i64, _i64, _P],
    "pxf_mask_from_index": [_P, _i64, _i64, _P, _P],
    "pxf_reduce_workspace_bytes": [],
    "pxf_moments": [_P] * 6 + [_i64, _P, _P, _P],
    "pxf_central_moments": [_P] * 3 + [_i64, _d, _d, _P, _P, _P],
    "pxf_transform_flat_moments": [_P] * 11 + [_i64] + [_d] * 7 + [_P, _P, _P],
    "pxf_rho": [_P, _P, _d, _d, _P, _i64, _P],
    "pxf_dist3": [_P, _P, _P, _d, _d, _d, _P, _i64, _P],
    "pxf_indangle": [_P] * 7 + [_i64, _i32, _P],
    "pxf_select_workspace_bytes": [],
    "pxf_kth_values": [_P, _i64, C.POINTER(C.c_int64), _i32, _P, _P, _P],
    "pxf_source": [_i32, _PP, _i64, _d, _d, _d, _d, _u64, _u64, _P],
    "pxf_source_ex": [_i32, _PP, _i64, _i64, _DP, _u64, _u64, _P],
    "pxf_select_passes": [],
    "pxf_select_state_layout": [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "pxf_select_begin": [C.POINTER(C.c_int64), _i32, _P, _P],
    "pxf_select_hist": [_P, _i64, _i32, _P, _i32, _P],
    "pxf_select_pick": [_i32, _P, _i32, _P, _P],
    "pxf_radial_kth_workspace_bytes": [_i64],
    "pxf_radial_kth": [_P, _P, _d, _d, _i32, _i64, C.POINTER(C.c_int64), _i32, _P, _P, _P, _P],
    "pxf_wselect_workspace_bytes": [],
    "pxf_wquantiles": [_P, _P, _i64, _DP, _i32, _P, _P, _P],
    "pxf_wselect_state_layout": [C.POINTER(C.c_int64)] * 4,
    "pxf_wselect_begin": [_DP, _i32, _P, _P],
    "pxf_wselect_hist": [_P, _P, _i64, _i32, _P, _i32, _P],
    "pxf_wselect_pick": [_i32, _P, _i32, _P],
    "pxf_wselect_neighbors": [_P, _i64, _i32, _P, _P],
    "pxf_wselect_finish": [_i32, _P, _P, _P],
    "pxf_griddata_workspace_bytes": [_i64],
    "pxf_griddata": [_P, _P, _P, _i64, _P, _i32, _i32, _d, _d, _d, _d, _DP, _P, _P, _i64, _i32, _P, _P, _P],
    "pxf_southwellbin_workspace_bytes": [_i32, _i32],
    "pxf_southwellbin": [_P, _P, _P, _P, _i64, _d, _P, _P, _P, _i32, _i32, _P, _P],
    "pxf_reconstruct": [_P, _P, _i32, _i32, _d, _d, _P, _P, _i32, _P, _P],
    "pxf_program_run": [_P, _i32, _PP, _i64, _P],
    "pxf_program_params": [_P, _i32, _DP, C.POINTER(C.c_int32)],
    "pxf_program_run_segments": [_P, _i32, _P, _i32, _i64, _PP, _i64, _P],
    "pxf_zernike_monomials": [_DP, _IP, _IP, _i32, _i32, _DP],
    "pxf_host_cos_atan2": [_DP, _DP, _DP, _DP, _DP, _i64],
    "pxf_host_sincos_atan2": [_DP, _DP, _DP, _DP, _DP, _i64],
    "pxf_fan_search": [_PP, _i64, _d, _d, _d, _d, _i32, _P, _P, _P, _P],
    "pxf_fan_finish": [_PP, _i64, _d, _d, _d, _d, _P, _i32, _P],
    "pxf_program_run_fast": [_P, _i32, _PP, _i64, _P],
    "pxf_program_run_segments_fast": [_P, _i32, _P, _i32, _i64, _PP, _i64, _P],
    "pxf_source_segments": [_i32, _PP, _i64, _P, _i32, _i64, _u64, _u64, _P],
    "pxf_program_plan": [_P, _i32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)],
}
_RESTYPE = {"pxf_strerror": C.c_char_p, "pxf_wolter_params": None,
            "pxf_compact_workspace_bytes": C.c_size_t, "pxf_reduce_workspace_bytes": C.c_size_t,
            "pxf_select_workspace_bytes": C.c_size_t, "pxf_southwellbin_workspace_bytes": C.c_size_t,
            "pxf_select_passes": C.c_int32, "pxf_select_state_layout": None,
            "pxf_wselect_workspace_bytes": C.c_size_t, "pxf_radial_kth_workspace_bytes": C.c_size_t, "pxf_wselect_state_layout": None,
            "pxf_griddata_workspace_bytes": C.c_size_t}

EXPORTS = tuple(_SIGS)
_lib = None


class PxfError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PxfError(
                "pyxfocus_amd: %s not found -- the HIP extension is not built "
                "(run __graft_entry__.build() or `make -C pyxfocus_amd/csrc`). "
                "There is no CPU fallback." % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, argtypes in _SIGS.items():
            f = getattr(h, name)      # AttributeError here = header/library mismatch: fail loudly
            f.argtypes = argtypes
            f.restype = _RESTYPE.get(name, C.c_int)
        if h.pxf_abi_version() != 1:
            raise PxfError("libpxf_hip.so ABI version mismatch")
        _lib = h
    return _lib


def check(code, what):
    if code != 0:
        h = lib()
        raise PxfError("%s failed: %s (code %d, hipError %d)" % (
            what, h.pxf_strerror(code).decode(), code, h.pxf_last_hip_error()))
