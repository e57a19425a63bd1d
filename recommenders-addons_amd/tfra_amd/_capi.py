"""ctypes binding of include/tfra_mi355x.h — the only door from Python to the HIP engine.

There is NO CPU fallback: if the shared library is missing or no MI355X is visible the
product path raises (TFRA's GPU ops likewise refuse to run without their .so,
R/tensorflow_recommenders_addons/utils/resource_loader.py:104-120).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtfra_mi355x.so")

TFRA_F32, TFRA_F16, TFRA_BF16, TFRA_I8, TFRA_I32, TFRA_I64, TFRA_F64 = range(7)
FLAG_UNIQUE_KEYS = 1
OPT_SGD, OPT_ADAM, OPT_ADAGRAD, OPT_FTRL = range(4)
OPTION_CAPTURE_SAFE = 1
OPTION_NO_OWNER_TAGS = 2
OPTION_KEY_BYTES_ON_DISK = 3


class TfraError(RuntimeError):
  def __init__(self, code, msg):
    super().__init__("tfra_mi355x error %d: %s" % (code, msg))
    self.code = code


class TableOpts(ctypes.Structure):
  _fields_ = [
      ("struct_size", ctypes.c_uint32),
      ("value_dtype", ctypes.c_int32),
      ("dim", ctypes.c_int32),
      ("aux_fields", ctypes.c_int32),
      ("init_capacity", ctypes.c_uint64),
      ("max_capacity", ctypes.c_uint64),
      ("max_hbm_for_vectors", ctypes.c_uint64),
      ("max_load_factor", ctypes.c_float),
      ("strategy", ctypes.c_int32),
      ("step_per_epoch", ctypes.c_int64),
      ("reserved_key_start_bit", ctypes.c_int32),
      ("device", ctypes.c_int32),
      ("aux_init", ctypes.c_float * 4),
  ]


class OptParams(ctypes.Structure):
  _fields_ = [
      ("kind", ctypes.c_int32),
      ("lr", ctypes.c_float),
      ("beta1", ctypes.c_float),
      ("beta2", ctypes.c_float),
      ("eps", ctypes.c_float),
      ("l1", ctypes.c_float),
      ("l2", ctypes.c_float),
      ("lr_power", ctypes.c_float),
      ("d_lr", ctypes.c_void_p),
  ]


_P = ctypes.c_void_p


class StepDesc(ctypes.Structure):
  """tfra_step_desc (include/tfra_mi355x.h): one table's step in tfra_multi_step_prefetch."""
  _fields_ = [
      ("struct_size", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("table", ctypes.c_void_p), ("opt", ctypes.c_void_p),
      ("plan_cur", ctypes.c_void_p), ("ids_cur", ctypes.c_void_p), ("rows_out", ctypes.c_void_p), ("find_default", ctypes.c_void_p),
      ("grads_or_values", ctypes.c_void_p), ("param_default_row", ctypes.c_void_p), ("scores", ctypes.c_void_p),
      ("plan_next", ctypes.c_void_p), ("ids_next", ctypes.c_void_p), ("n_next", ctypes.c_size_t), ("main_stream", ctypes.c_void_p),
      ("side_stream", ctypes.c_void_p),
  ]


class OverlapStep(ctypes.Structure):
  """tfra_overlap_step (include/tfra_mi355x.h): one step of tfra_table_steps_overlap."""
  _fields_ = [
      ("struct_size", ctypes.c_uint32), ("default_is_full", ctypes.c_int32), ("n", ctypes.c_size_t), ("ids", ctypes.c_void_p),
      ("rows_out", ctypes.c_void_p), ("exists_out", ctypes.c_void_p), ("defaults", ctypes.c_void_p), ("values_prev", ctypes.c_void_p),
      ("scores_prev", ctypes.c_void_p), ("n_next", ctypes.c_size_t), ("ids_next", ctypes.c_void_p),
      ("n_next2", ctypes.c_size_t), ("ids_next2", ctypes.c_void_p),
  ]


ALLTOALLV_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t),
                                ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p)


class Transport(ctypes.Structure):
  """tfra_transport (include/tfra_mi355x.h): the alltoallv the routed step driver exchanges buffers through."""
  _fields_ = [("ctx", ctypes.c_void_p), ("rank", ctypes.c_int), ("world", ctypes.c_int), ("alltoallv", ALLTOALLV_FN),
              ("alltoallv2", ctypes.c_void_p)]   # optional: NULL for the host-staged test transport


RCCL_ID_BYTES = 128
ROUTE_NO_THREAD = 1
_SZ = ctypes.c_size_t
_I = ctypes.c_int
_SIGS = {
    "tfra_table_create": [ctypes.POINTER(TableOpts), _P, ctypes.POINTER(_P)],
    "tfra_table_destroy": [_P],
    "tfra_table_find": [_P, _SZ, _P, _P, _P, _P, _I, _P],
    "tfra_table_find_field": [_P, _I, _SZ, _P, _P, _P, _P, _I, _P],
    "tfra_table_insert_or_assign": [_P, _SZ, _P, _P, _P, ctypes.c_uint32, _P],
    "tfra_table_find_n": [_P, _SZ, _P, _P, _P, _P, _P, ctypes.c_int, _P],
    "tfra_table_insert_or_assign_n": [_P, _SZ, _P, _P, _P, _P, _P],
    "tfra_table_insert_field": [_P, _I, _SZ, _P, _P, ctypes.c_uint32, _P],
    "tfra_table_accum_or_assign": [_P, _SZ, _P, _P, _P, _P, ctypes.c_uint32, _P],
    "tfra_table_erase": [_P, _SZ, _P, _P],
    "tfra_table_clear": [_P, _P],
    "tfra_table_size": [_P, ctypes.POINTER(_SZ), _P],
    "tfra_table_size_to_device": [_P, _P, _P],
    "tfra_table_capacity": [_P, ctypes.POINTER(_SZ)],
    "tfra_table_growth_stats": [_P, ctypes.POINTER(ctypes.c_uint64)],
    "tfra_multi_step_prefetch": [_SZ, _P, _I],
    "tfra_plan_reduce_to": [_P, _P, _P, _P, _P],
    "tfra_table_check_errors": [_P, _P],
    "tfra_table_slot_census": [_P, ctypes.POINTER(ctypes.c_uint64), _P],
    "tfra_table_reserve": [_P, _SZ, _P],
    "tfra_table_export_batch": [_P, _SZ, _SZ, _P, _P, _P, _P, _P],
    "tfra_table_set_global_epoch": [_P, ctypes.c_uint64],
    "tfra_table_set_option": [_P, _I, ctypes.c_int64],
    "tfra_table_save": [_P, ctypes.c_char_p, _SZ, _I, _P, ctypes.POINTER(_SZ)],
    "tfra_table_load": [_P, ctypes.c_char_p, _SZ, _P, ctypes.POINTER(_SZ)],
    "tfra_table_save_field": [_P, _I, ctypes.c_char_p, _SZ, _I, _P, ctypes.POINTER(_SZ)],
    "tfra_table_load_field": [_P, _I, ctypes.c_char_p, _SZ, _P, ctypes.POINTER(_SZ)],
    "tfra_table_apply_optimizer": [_P, ctypes.POINTER(OptParams), _SZ, _P, _P, _P, _I, _P, _P],
    "tfra_table_apply_sparse": [_P, ctypes.POINTER(OptParams), _SZ, _P, _P, _P, _P],
    "tfra_sparse_plan_create": [_I, _P],
    "tfra_sparse_plan_destroy": [_P],
    "tfra_sparse_plan_build": [_P, _SZ, _P, _I, _P],
    "tfra_table_apply_planned": [_P, ctypes.POINTER(OptParams), _P, _P, _P, _P],
    "tfra_table_step_prefetch": [_P, ctypes.POINTER(OptParams), _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _P],
    "tfra_table_step_prefetch_assign": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _P],
    "tfra_step_driver_create": [_P, ctypes.POINTER(_P)],
    "tfra_step_driver_destroy": [_P],
    "tfra_table_step_overlap": [_P, _SZ, _P, _P, _P, _P, _I, _P, _P, _SZ, _P, _SZ, _P, _P],
    "tfra_table_step_overlap_flush": [_P, _P, _P, _P],
    "tfra_table_steps_overlap": [_P, _SZ, _P, _P],
    "tfra_step_driver_timing": [_P, _P],
    "tfra_step_driver_lookups_listed": [_P, ctypes.POINTER(ctypes.c_uint64)],
    "tfra_step_driver_time_kernels": [_P, _SZ],
    "tfra_step_driver_kernel_times": [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_SZ)],
    "tfra_step_driver_stats": [_P, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(_I), _P, _P, _P],
    "tfra_table_upsert_sparse": [_P, _SZ, _P, _P, _P, _P],
    "tfra_table_upsert_planned": [_P, _P, _P, _P, _P],
    "tfra_sparse_plan_read": [_P, _P, _P, _P, _P, _SZ, _P],
    "tfra_workspace_create": [_I, ctypes.POINTER(_P)],
    "tfra_workspace_destroy": [_P],
    "tfra_unique": [_P, _SZ, _P, _P, _P, _P, _P],
    "tfra_unique_unordered": [_P, _SZ, _P, _P, _P, _P, _P],
    "tfra_table_find_unique": [_P, _P, _SZ, _P, _P, _P, _P, _I, _P, _P, _P, _P],
    "tfra_segment_sum": [_P, _SZ, _I, _P, _P, _P, _SZ, _P, _P],
    "tfra_gather_rows": [_SZ, _SZ, _P, _P, _P, _P],
    "tfra_keys_widen_i32": [_SZ, _P, _P, _P],
    "tfra_keys_narrow_i32": [_SZ, _P, _P, _P, _P],
    "tfra_sparse_segment_combine": [_P, _SZ, _I, _P, _P, _P, _P, _I, _SZ, _P, _P],
    "tfra_partition": [_P, _SZ, _P, _P, _I, _I, _P, _P, _P, _P],
    "tfra_partition_by_owner": [_P, _SZ, _P, _I, _P, _P, _P],
    "tfra_scatter_rows": [_SZ, _SZ, _P, _P, _P, _P],
    "tfra_reduce_by_key": [_P, _SZ, _P, _I, _P, _P, _P, _P, _P],
    "tfra_select_lowest": [_P, _SZ, _P, _P, _I, _SZ, _P, _P],
    "tfra_plan_partition": [_P, _P, _I, _I, _P, _P, _P, _P],
    "tfra_plan_positions_to": [_P, _P, _P, _P],
    "tfra_rccl_unique_id": [ctypes.c_char_p, _P],
    "tfra_rccl_transport_create": [ctypes.c_char_p, _P, _I, _I, _I, ctypes.POINTER(Transport)],
    "tfra_rccl_transport_destroy": [ctypes.POINTER(Transport)],
    "tfra_rccl_transport_ranks": [ctypes.POINTER(Transport), ctypes.POINTER(_I)],
    "tfra_route_create": [_P, ctypes.POINTER(Transport), _I, _SZ, ctypes.c_uint32, ctypes.POINTER(_P)],
    "tfra_route_destroy": [_P],
    "tfra_route_feed": [_P, _SZ, _P, _I, _P],
    "tfra_route_lookup": [_P, _P, _P, _P],
    "tfra_route_apply": [_P, ctypes.POINTER(OptParams), _P, _P, _P],
    "tfra_route_served_ids": [_P, ctypes.POINTER(_P), ctypes.POINTER(_SZ), ctypes.POINTER(_SZ)],
    "tfra_assign_route_create": [_P, ctypes.POINTER(Transport), _I, _SZ, ctypes.POINTER(_P)],
    "tfra_assign_route_destroy": [_P],
    "tfra_assign_route_feed": [_P, _SZ, _P, _I, _P],
    "tfra_assign_route_step": [_P, _P, _P, _P, _P],
    "tfra_assign_route_flush": [_P, _P, _P],
    "tfra_assign_route_stats": [_P, ctypes.POINTER(ctypes.c_uint64)],
    "tfra_assign_route_time_kernels": [_P, _SZ],
    "tfra_assign_route_kernel_times": [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_SZ)],
}

_lib = None


def lib():
  """Loads the HIP library (fails loudly when it was not built)."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise ImportError(
          "%s not found: run `python recommenders-addons_amd/build.py` (hipcc, gfx950). "
          "There is no CPU fallback." % LIB_PATH)
    l = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
      f = getattr(l, name)
      f.argtypes = args
      f.restype = ctypes.c_int
    l.tfra_last_error.restype = ctypes.c_char_p
    l.tfra_last_error.argtypes = []
    l.tfra_abi_version.restype = ctypes.c_int
    l.tfra_abi_version.argtypes = []
    _lib = l
  return _lib


def check(rc):
  if rc != 0:
    raise TfraError(rc, lib().tfra_last_error().decode("utf-8", "replace"))


def call(name, *args):
  check(getattr(lib(), name)(*args))
