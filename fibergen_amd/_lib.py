"""ctypes binding of libfibergen_amd.so (the C ABI declared in include/fibergen_amd.h).

There is no CPU fallback: if the HIP library is missing or no GPU is present the
product path fails loudly.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfibergen_amd.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
CALLBACK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)
LOADSTEP_CALLBACK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int)



class FgFiber(ctypes.Structure):
    """struct fg_fiber of include/fibergen_amd.h"""
    _fields_ = [("kind", ctypes.c_int), ("material", ctypes.c_int), ("c", ctypes.c_double * 3),
                ("a", ctypes.c_double * 3), ("L", ctypes.c_double), ("R", ctypes.c_double)]


class FgXop(ctypes.Structure):
    """struct fg_xop: one send / receive of an exchange handed to the callback transport"""
    _fields_ = [("send", ctypes.c_int), ("peer", ctypes.c_int), ("ptr", ctypes.c_void_p), ("bytes", ctypes.c_ulong)]


class FgPlanOp(ctypes.Structure):
    """struct fg_plan_op: one entry of the exchange plan (offsets / counts in doubles)"""
    _fields_ = [("send", ctypes.c_int), ("peer", ctypes.c_int), ("buffer", ctypes.c_int), ("offset", ctypes.c_long),
                ("count", ctypes.c_long)]


EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(FgXop), ctypes.c_int)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_int)

# name -> (restype, argtypes); mirrors include/fibergen_amd.h one to one
SIGNATURES = {
    "fg_abi_version": (ctypes.c_int, []),
    "fg_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "fg_create": (ctypes.c_void_p, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_double, ctypes.c_int]),
    "fg_destroy": (None, [ctypes.c_void_p]),
    "fg_set_num_phases": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "fg_set_phase": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, c_double_p]),
    "fg_set_normals": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "fg_set_option_d": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]),
    "fg_set_option_i": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_long]),
    "fg_set_bc_projector": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "fg_set_convergence_callback": (ctypes.c_int, [ctypes.c_void_p, CALLBACK, ctypes.c_void_p]),
    "fg_cancel": (ctypes.c_int, [ctypes.c_void_p]),
    "fg_run_load_case": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, ctypes.POINTER(ctypes.c_int)]),
    "fg_run_load_steps": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                         LOADSTEP_CALLBACK, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "fg_iterate": (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int]),
    "fg_time_iterations": (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int, c_double_p]),
    "fg_get_iterations": (ctypes.c_long, [ctypes.c_void_p]),
    "fg_get_residuals": (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int]),
    "fg_get_solve_time": (ctypes.c_double, [ctypes.c_void_p]),
    "fg_mean_stress": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "fg_mean_strain": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "fg_volume_fraction": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p]),
    "fg_calc_ref_material": (ctypes.c_int, [ctypes.c_void_p]),
    "fg_get_ref_material": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    "fg_field_components": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "fg_get_field": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_double_p]),
    "fg_set_field": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_double_p]),
    "fg_device_pointer": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]),
    "fg_get_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    "fg_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "fg_run_stage": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p]),
    "fg_enable_stage_timing": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "fg_get_stage_times": (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.POINTER(ctypes.c_long)]),
    "fg_get_stage_timing_bias": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "fg_get_counter": (ctypes.c_long, [ctypes.c_void_p, ctypes.c_char_p]),
    "fg_get_comm_times": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "fg_device_pci_bus_id": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    "fg_hbm_stream": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p]),
    "fg_create_slab": (ctypes.c_void_p, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "fg_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "fg_slab_connect_rccl": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "fg_slab_group_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                            ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "fg_slab_connect_callback": (ctypes.c_int, [ctypes.c_void_p, EXCHANGE_FN, ALLREDUCE_FN, ctypes.c_void_p]),
    "fg_slab_transport": (ctypes.c_char_p, [ctypes.c_void_p]),
    "fg_slab_plan": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, ctypes.POINTER(FgPlanOp), ctypes.c_int, ctypes.POINTER(FgPlanOp)]),
    "fg_voxelize": (ctypes.c_int, [ctypes.POINTER(FgFiber), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_double, ctypes.c_double, ctypes.c_double, c_double_p, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p, c_double_p, c_double_p,
                                   ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    "fg_voxelize_team_depth": (ctypes.c_int, [ctypes.c_int]),
}

_lib = None


def load():
    """Load the HIP library; raises RuntimeError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "fibergen_amd: %s not found. Build it with `make -C fibergen_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.fg_abi_version() != 1:
        raise RuntimeError("fibergen_amd: ABI version mismatch")
    _lib = lib
    return lib
