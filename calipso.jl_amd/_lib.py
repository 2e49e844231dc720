"""ctypes loader for libcalipso_hip.so (the C ABI of include/calipso_hip.h).

There is no CPU path: if the library is missing or no HIP device is usable, everything raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CALIPSO_HIP_LIB") or os.path.join(_HERE, "libcalipso_hip.so")   # (CALIPSO_HIP_LIB: another build of the same library, for A/B timing)
_LIB = None

EVAL_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                      C.POINTER(C.c_double))

CALLBACK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
DEVICE_EVAL_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)

# every symbol include/calipso_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _u32, _u64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
_pd, _pi64, _pi32 = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
SYMBOLS = {
    "calipso_hip_create": (_i32, [_i64, _i64, _i64, _i64, _i64, _pi64, _i64, _pi64, _pi64, _i32, C.POINTER(_vp)]),
    "calipso_hip_create_structured": (_i32, [_i64, _i64, _i64, _i64, _i64, _pi64, _i64, _pi64, _pi64, _i32, _pi64, _pi64, _i64, _pi64, C.POINTER(_vp)]),
    "calipso_hip_destroy": (_i32, [_vp]),
    "calipso_hip_last_error": (C.c_char_p, [_vp]),
    "calipso_hip_version": (C.c_char_p, []),
    "calipso_hip_device_count": (_i32, []),
    "calipso_hip_set_field": (_i32, [_vp, C.c_char_p, _pd, _i64]),
    "calipso_hip_get_field": (_i32, [_vp, C.c_char_p, _pd, _i64]),
    "calipso_hip_set_sparsity": (_i32, [_vp, C.c_char_p, _i64, _pi64, _pi64]),
    "calipso_hip_scatter_field": (_i32, [_vp, C.c_char_p, _pd, _i64]),
    "calipso_hip_scatter_hessian": (_i32, [_vp, _pd, _i64, _pd, _i64, _pd, _i64]),
    "calipso_hip_get_index": (_i64, [_vp, C.c_char_p, _pi64, _i64]),
    "calipso_hip_cone": (_i32, [_vp, _i32, _i32]),
    "calipso_hip_residual": (_i32, [_vp]),
    "calipso_hip_violations": (_i32, [_vp, _pd]),
    "calipso_hip_residual_jacobian_variables_symmetric": (_i32, [_vp]),
    "calipso_hip_jacobian_variables_mul": (_i32, [_vp, _pd, _pd]),
    "calipso_hip_factorize": (_i32, [_vp, _pi64]),
    "calipso_hip_inertia_correction": (_i32, [_vp, _pi64]),
    "calipso_hip_residual_symmetric": (_i32, [_vp, _i32]),
    "calipso_hip_linear_solve": (_i32, [_vp]),
    "calipso_hip_search_direction_symmetric": (_i32, [_vp, _i32]),
    "calipso_hip_iterative_refinement": (_i32, [_vp, _pi32, _pd]),
    "calipso_hip_search_direction": (_i32, [_vp]),
    "calipso_hip_search_direction_nonsymmetric": (_i32, [_vp]),
    "calipso_hip_cone_search": (_i32, [_vp, _pd, _pd]),
    "calipso_hip_cone_violation": (_i32, [_vp, _pd, _pd, _dbl, _pi32]),
    "calipso_hip_candidate": (_i32, [_vp, _dbl, _i32]),
    "calipso_hip_merit": (_i32, [_vp, _i32, _pd]),
    "calipso_hip_merit_gradient": (_i32, [_vp]),
    "calipso_hip_constraint_violation": (_i32, [_vp, _i32, _pd]),
    "calipso_hip_merit_directional": (_i32, [_vp, _pd]),
    "calipso_hip_accept": (_i32, [_vp, _dbl]),
    "calipso_hip_initialize": (_i32, [_vp, _pd]),
    "calipso_hip_solve": (_i32, [_vp, EVAL_FN, _vp]),
    "calipso_hip_differentiate": (_i32, [_vp, EVAL_FN, _vp]),
    "calipso_hip_set_device_evaluator": (_i32, [_vp, C.c_void_p, _vp]),
    "calipso_hip_set_device_block_evaluator": (_i32, [_vp, C.c_void_p, _vp]),
    "calipso_hip_device_evaluate": (_i32, [_vp, _i32, _u32]),
    "calipso_hip_set_callbacks": (_i32, [_vp, C.c_void_p, C.c_void_p, _vp]),
    "calipso_hip_stats": (_i32, [_vp, _pi64]),
    "calipso_hip_qp_attach": (_i32, [_vp, _pd, _pd, _pd, _pd, _pd, _pd, _dbl]),
    "calipso_hip_qp_evaluate": (_i32, [_vp, _i32, _u32]),
    "calipso_hip_newton_step": (_i32, [_vp, _i32, _pd]),
    "calipso_hip_newton_steps": (_i32, [_vp, _i32, _i32, _pd, _pi32]),
    "calipso_hip_phase_times": (_i32, [_vp, _pd]),
    "calipso_hip_kernel_times": (_i32, [_vp, _pd]),
    "calipso_hip_structure_work": (_i32, [_vp, _pd]),
    "calipso_hip_analyze_structure": (_i32, [_vp, _pi64]),
    "calipso_hip_clear_structure": (_i32, [_vp]),
    "calipso_hip_set_stage_parallel": (_i32, [_vp, _i32, _i32, _pi64]),
    "calipso_hip_set_stage_blocks": (_i32, [_vp, _i32, _pi64]),
    "calipso_hip_group_create": (_i32, [C.POINTER(_vp), _i32, C.POINTER(_vp)]),
    "calipso_hip_group_destroy": (_i32, [_vp]),
    "calipso_hip_group_newton_step": (_i32, [_vp, _i32, _pd, C.POINTER(_i32)]),
    "calipso_hip_group_solve": (_i32, [_vp, C.POINTER(_i32)]),
    "calipso_hip_group_set_evaluators": (_i32, [_vp, C.POINTER(EVAL_FN), C.POINTER(_vp)]),
    "calipso_hip_ldl_create": (_i32, [_i64, _i32, C.POINTER(_vp)]),
    "calipso_hip_ldl_factorize_csc": (_i32, [_vp, _i64, _pi64, _pi64, _pd, _pi64]),
    "calipso_hip_ldl_inertia": (_i32, [_vp, _pi64]),
    "calipso_hip_ordering": (_i32, [_i64, _pi64, _pi64, _i32, _pi64]),
    "calipso_hip_symbolic": (_i64, [_i64, _pi64, _pi64, _pi64, _pi64, _pi64, _pi64, _pi64, _pi64, _pi64]),
    "calipso_hip_ldl_analyze_csc": (_i32, [_vp, _i64, _pi64, _pi64, _i32, _pi64, _pi64]),
    "calipso_hip_ldl_solve": (_i32, [_vp, _i64, _i64, _pd, _pd]),
    "calipso_hip_sparse_create": (_i32, [_i64, _pi64, _pi64, _i32, _pi64, _i32, C.POINTER(_vp)]),
    "calipso_hip_sparse_destroy": (_i32, [_vp]),
    "calipso_hip_sparse_last_error": (C.c_char_p, [_vp]),
    "calipso_hip_sparse_info": (_i32, [_vp, _pi64]),
    "calipso_hip_sparse_set_batch": (_i32, [_vp, _i64]),
    "calipso_hip_sparse_select": (_i32, [_vp, _i64]),
    "calipso_hip_sparse_factorize": (_i32, [_vp, _pd, _pi64]),
    "calipso_hip_sparse_solve": (_i32, [_vp, _i64, _pd, _pd]),
    "calipso_hip_sparse_factorize_device": (_i32, [_vp, _vp, _pi64]),
    "calipso_hip_sparse_solve_device": (_i32, [_vp, _i64, _vp, _vp]),
    "calipso_hip_sparse_get_factor": (_i32, [_vp, _pi64, _pi64, _pi64, _pd, _pd]),
    "calipso_hip_sparse_timing": (_i32, [_vp, _pd]),
    "calipso_hip_small_create": (_i32, [_i64, _i64, _i64, _i32, C.POINTER(_vp)]),
    "calipso_hip_small_destroy": (_i32, [_vp]),
    "calipso_hip_small_last_error": (C.c_char_p, [_vp]),
    "calipso_hip_small_set": (_i32, [_vp, _pd, _pd]),
    "calipso_hip_small_solve": (_i32, [_vp, _pd]),
    "calipso_hip_small_get": (_i32, [_vp, _pd, _pi64]),
    "calipso_hip_smallnewton_create": (_i32, [_i64, _i64, _i64, _i64, _i32, C.POINTER(_vp)]),
    "calipso_hip_smallnewton_destroy": (_i32, [_vp]),
    "calipso_hip_smallnewton_last_error": (C.c_char_p, [_vp]),
    "calipso_hip_smallnewton_set_option": (_i32, [_vp, C.c_char_p, _dbl]),
    "calipso_hip_smallnewton_set_cones": (_i32, [_vp, _i64, _i64, _pi64]),
    "calipso_hip_smallnewton_set_qp": (_i32, [_vp, _pd, _pd, _pd, _pd, _pd, _pd, _dbl, _i32]),
    "calipso_hip_smallnewton_set_state": (_i32, [_vp, _pd, _pd, _pd]),
    "calipso_hip_smallnewton_get_state": (_i32, [_vp, _pd, _pd, _pd, _pi64]),
    "calipso_hip_smallnewton_trace": (_i32, [_vp, _i32, _pd]),
    "calipso_hip_smallnewton_solve": (_i32, [_vp, _pi32, _pd]),
    "calipso_hip_smallnewton_steps": (_i32, [_vp, _i32, _i32, _pd, _pi32, _pd]),
    "calipso_hip_smallnewton_differentiate": (_i32, [_vp, _i64, _i32, _pd, _pd, _pi32, _pd]),
    "calipso_hip_comm_unique_id": (_i32, [C.POINTER(C.c_uint8)]),
    "calipso_hip_comm_init": (_i32, [_i32, _i32, C.POINTER(C.c_uint8), _i32, C.POINTER(_vp)]),
    "calipso_hip_comm_destroy": (_i32, [_vp]),
    "calipso_hip_comm_size": (_i32, [_vp, _pi32]),
    "calipso_hip_comm_last_error": (C.c_char_p, [_vp]),
    "calipso_hip_comm_gather_status": (_i64, [_vp, _pi32, _i64, _pi32, _i64, _pi64]),
    "calipso_hip_comm_allreduce_sum": (_i32, [_vp, _pd, _i64]),
    "calipso_hip_mfma_f64_peak": (_i32, [_i32, _pd]),
    "calipso_hip_synchronize": (_i32, [_vp]),
    "calipso_hip_streams_concurrent": (_i32, [_vp, _vp, _pd]),
    "calipso_hip_rebind_stream": (_i32, [_vp, _i32]),
    "calipso_hip_splitmix_uniform": (_i32, [_u64, _u64, _dbl, _dbl, _i64, _pd]),
}


class CalipsoHipError(RuntimeError):
    pass


def lib():
    """Load the library (once).  Raises if it has not been built: the product never falls back to a CPU path."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise CalipsoHipError("libcalipso_hip.so not found at %s — build it with `python __graft_entry__.py build` "
                                  "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)   # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB
