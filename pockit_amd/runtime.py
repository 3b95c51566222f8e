"""ctypes binding of libpockit_hip.so (the C ABI of include/pockit_hip.h).

Thin by design: structures, prototypes and error translation only.  A missing library, a missing
GPU or a failing HIP call raises ``RuntimeError`` -- the package has no CPU evaluation path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import hipbuild

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_phase", "n_I", "nred", "lds_g", "lds_j", "lds_h", "ne_j", "ne_h", "prepass_f", "prepass_grad",
                 "prepass_g", "prepass_jac", "prepass_hess", "lds_x", "ne_a", "ne_hc", "lds_e", "tab_cap", "sharded",
                 "lds_jc", "ne_jc", "max_phases", "cycle_subs", "hess_subs", "hessc_subs", "jacc_subs", "big_global", "big_rows", "wide")]


class ProblemDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("n_sys", C.c_int32), ("n_s", C.c_int32), ("l_s", C.c_int32),
        ("n_phase", C.c_int32), ("n_tiles", C.c_int32), ("n_kinds", C.c_int32),
        ("nnz_J", C.c_int64), ("nnz_H", C.c_int64),
        ("phases", C.c_void_p), ("tiles", C.c_void_p), ("kinds", C.c_void_p),
        ("items_jac", C.c_void_p), ("n_items_jac", C.c_int32),
        ("items_hess", C.c_void_p), ("n_items_hess", C.c_int32),
        ("ib", c_int32_p), ("n_ib", C.c_int64),
        ("db", c_double_p), ("n_db", C.c_int64),
        ("lb", C.POINTER(C.c_int64)), ("n_lb", C.c_int64),
        ("gz_off", C.c_int32), ("n_gz", C.c_int32),
        ("items_aux", C.c_void_p), ("n_items_aux", C.c_int32),
        ("outer", C.c_void_p), ("n_outer", C.c_int32), ("n_aux", C.c_int32),
        ("items_hessc", C.c_void_p), ("n_items_hessc", C.c_int32), ("nnz_Hc", C.c_int64),
        ("jac_row", c_int32_p), ("jac_col", c_int32_p), ("hess_row", c_int32_p), ("hess_col", c_int32_p),
        ("items_jacc", C.c_void_p), ("n_items_jacc", C.c_int32), ("nnz_Jc", C.c_int64),
    ]


# numpy mirrors of csrc/pk_abi.h
PHASE_FIELDS = ["scheme", "n_x", "n_u", "n_c", "L_m", "L_d", "state_len", "L", "x_off", "g_off", "path_off",
                "mid_lo", "mid_hi", "tile_lo", "tile_hi", "tau_off", "w_off", "width_off", "jseg_off", "jt_off",
                "hseg_off", "red_off", "aseg_off", "hcseg_off", "ivK_off", "ivfull_off", "ivld_off", "n_int", "jcseg_off", "jct_off"]
PHASE_DTYPE = np.dtype([(n, np.int32) for n in PHASE_FIELDS])
TILE_FIELDS = ["phase", "j0", "nj", "kid", "kidf", "q0", "r0", "offI", "offT", "K", "last",
               "nnzI", "nnzT", "irc_off", "iv_off", "tv_off", "full_off", "pad", "magicI", "magicR", "magicT", "stage"]
TILE_DTYPE = np.dtype([(n, np.uint32 if n.startswith("magic") else np.int32) for n in TILE_FIELDS])
KIND_FIELDS = ["K", "R", "nnzI", "nnzT", "irc_off", "iv_off", "tv_off", "full_off"]
KIND_DTYPE = np.dtype([(n, np.int32) for n in KIND_FIELDS])
ITEM_DTYPE = np.dtype([("pos", np.int64), ("coef", np.float64), ("eid", np.int32), ("lam", np.int32)])
OUTER_DTYPE = np.dtype([("pos", np.int64), ("offA", np.int32), ("lenA", np.int32), ("offB", np.int32), ("lenB", np.int32),
                        ("offM", np.int32), ("flags", np.int32), ("count", np.int32), ("pad", np.int32)])
ERRIV_DTYPE = np.dtype([("phase", np.int32), ("K", np.int32), ("lm", np.int32), ("row0", np.int32),
                        ("tab_off", np.int32), ("tau_off", np.int32), ("rows", np.int32), ("stage", np.int32),
                        ("out_off", np.int64), ("width", np.float64)])

WAVES_PER_BLOCK = int(os.environ.get("POCKIT_AMD_WPB") or 4)  # PK_WAVES_PER_BLOCK of csrc/pk_abi.h (POCKIT_AMD_WPB: experiments)
WAVE = 64  # PK_WAVE
KERNELS = ["pk_int", "pk_fin", "pk_g", "pk_grad", "pk_jac", "pk_hess", "pk_xall", "pk_aux", "pk_outer", "pk_hessc", "pk_err", "pk_csr",
           "pk_cycle", "pk_xchg", "pk_runs", "pk_jacc", "pk_cyclec"]
EXPORTS = ["pk_create", "pk_destroy", "pk_last_error", "pk_device_count", "pk_load_model", "pk_set_problem",
           "pk_get_structure", "pk_eval_f", "pk_eval_grad", "pk_eval_g", "pk_eval_jac", "pk_eval_hess",
           "pk_eval_f_dev", "pk_eval_grad_dev", "pk_eval_g_dev", "pk_eval_jac_dev", "pk_eval_hess_dev",
           "pk_eval_cycle_dev", "pk_eval_cycle_dev_repeat", "pk_eval_hessc_prepared", "pk_sync", "pk_profile", "pk_profile_read", "pk_kernel_name",
           "pk_set_shard", "pk_eval_integrals_dev", "pk_eval_f_from_integrals_dev", "pk_aux_buffer", "pk_eval_outer_dev", "pk_store_word_dev", "pk_callback_cycle", "pk_eval_cycle",
           "pk_prepare_x", "pk_fetch", "pk_eval_hess_prepared", "pk_host_buffer", "pk_eval_hessc", "pk_eval_hessc_dev",
           "pk_set_mesh_error_tables", "pk_eval_mesh_error", "pk_eval_mesh_error_dev", "pk_set_cycle_graph", "pk_profile_sampling",
           "pk_set_csr_map", "pk_gather_csr_dev", "pk_eval_jac_csr_dev", "pk_eval_hess_csr_dev", "pk_eval_jac_csr",
           "pk_eval_hess_csr", "pk_trace_read", "pk_set_cycle_mode", "pk_same_x", "pk_set_result_targets",
           "pk_result_location", "pk_set_host_mode", "pk_stage_lambda", "pk_invalidate_x", "pk_host_alloc", "pk_host_free",
           "pk_device_alloc", "pk_device_free", "pk_ipc_export", "pk_ipc_open", "pk_ipc_close", "pk_set_shared_grad_target",
           "pk_set_exchange", "pk_exchange_sums_dev", "pk_copy_runs_dev", "pk_set_exchange_inline",
           "pk_host_register", "pk_host_unregister", "pk_copy_dev", "pk_eval_xpart_dev",
           "pk_eval_jacc", "pk_eval_jacc_dev", "pk_callback_x", "pk_callback_hess", "pk_set_jac_constant_runs", "pk_fill_jac_constants", "pk_set_host_option",
           "pk_set_jacobian_layout", "pk_exchange_status", "pk_wait_idle", "pk_same_bits", "pk_copy_bits", "pk_host_threads", "pk_host_threads_hot", "pk_host_threads_jobs", "pk_set_cycle_layout"]

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as
    the system one); if our library pulled in /opt/rocm's copy first, a later ``import torch`` would bind to
    it and fail to find the GPU ("No HIP GPUs are available").  So when torch is installed, load ITS runtime
    first (without importing torch); our library's DT_NEEDED libamdhip64.so.7 then resolves to it by SONAME."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """Load (building if the sources are newer) libpockit_hip.so and declare prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    path = hipbuild.LIB_PATH
    if not os.path.exists(path):
        path = hipbuild.build_runtime()
    _preload_hip_runtime()
    try:
        lib = C.CDLL(path)
    except OSError as exc:
        raise RuntimeError(f"cannot load {path}: {exc}; the MI355X evaluator has no CPU fallback") from exc
    vp, dp = C.c_void_p, c_double_p
    lib.pk_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.pk_destroy.argtypes = [vp]
    lib.pk_destroy.restype = None
    lib.pk_last_error.argtypes = [vp]
    lib.pk_last_error.restype = C.c_char_p
    lib.pk_device_count.restype = C.c_int
    lib.pk_load_model.argtypes = [vp, vp, C.c_size_t, C.POINTER(ModelDesc)]
    lib.pk_set_problem.argtypes = [vp, C.POINTER(ProblemDesc)]
    lib.pk_get_structure.argtypes = [vp, c_int32_p, c_int32_p, c_int32_p, c_int32_p]
    lib.pk_eval_f.argtypes = [vp, dp, dp]
    lib.pk_eval_grad.argtypes = [vp, dp, dp]
    lib.pk_eval_g.argtypes = [vp, dp, dp]
    lib.pk_eval_jac.argtypes = [vp, dp, dp]
    lib.pk_eval_hess.argtypes = [vp, dp, dp, C.c_double, dp]
    lib.pk_eval_cycle.argtypes = [vp, dp, dp, C.c_double, dp, dp, dp, dp, dp]
    lib.pk_prepare_x.argtypes = [vp, dp]
    lib.pk_fetch.argtypes = [vp, C.c_int, dp]
    lib.pk_eval_hess_prepared.argtypes = [vp, dp, C.c_double, dp]
    lib.pk_host_buffer.argtypes = [vp, C.c_int, C.POINTER(dp), C.POINTER(C.c_int64)]
    lib.pk_same_x.argtypes = [vp, dp]
    lib.pk_stage_lambda.argtypes = [vp, dp]
    lib.pk_set_result_targets.argtypes = [vp, dp, dp, dp, dp, dp]
    lib.pk_result_location.argtypes = [vp, C.c_int, C.POINTER(dp)]
    lib.pk_set_host_mode.argtypes = [vp, C.c_int, C.c_int]
    lib.pk_invalidate_x.argtypes = [vp]
    lib.pk_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.pk_host_free.argtypes = [vp]
    lib.pk_device_alloc.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.pk_device_free.argtypes = [vp, vp]
    lib.pk_ipc_export.argtypes = [vp, vp, vp]
    lib.pk_ipc_open.argtypes = [vp, vp, C.POINTER(vp)]
    lib.pk_ipc_close.argtypes = [vp, vp]
    lib.pk_set_shared_grad_target.argtypes = [vp, vp]
    lib.pk_set_exchange.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int]
    lib.pk_exchange_sums_dev.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.pk_copy_runs_dev.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    lib.pk_set_exchange_inline.argtypes = [vp, C.c_int]
    lib.pk_host_register.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp)]
    lib.pk_host_unregister.argtypes = [vp, vp]
    lib.pk_copy_dev.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.pk_eval_xpart_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    # (raw addresses on the per-callback entry points: building a typed pointer costs more than the call)
    lib.pk_callback_x.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    lib.pk_callback_hess.argtypes = [vp, vp, vp, C.c_double, vp, vp, C.c_int, vp]
    lib.pk_callback_cycle.argtypes = [vp, vp, vp, C.c_double, vp, vp, vp]
    lib.pk_set_jac_constant_runs.argtypes = [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.pk_fill_jac_constants.argtypes = [vp, vp]
    lib.pk_set_host_option.argtypes = [vp, C.c_char_p, C.c_int]
    lib.pk_set_jacobian_layout.argtypes = [vp, C.c_int]
    lib.pk_exchange_status.argtypes = [vp, vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.pk_eval_hessc.argtypes = [vp, dp, dp, C.c_double, dp]
    lib.pk_eval_jacc.argtypes = [vp, dp, dp]
    lib.pk_eval_jacc_dev.argtypes = [vp, vp, vp, vp]
    lib.pk_eval_hessc_dev.argtypes = [vp, vp, vp, C.c_double, vp, vp]
    lib.pk_set_mesh_error_tables.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, dp, C.c_int64, C.c_int64]
    lib.pk_eval_mesh_error.argtypes = [vp, dp, dp, dp]
    lib.pk_eval_mesh_error_dev.argtypes = [vp, vp, vp, vp, vp]
    lib.pk_set_cycle_graph.argtypes = [vp, C.c_int]
    lib.pk_set_cycle_mode.argtypes = [vp, C.c_int]
    lib.pk_set_cycle_layout.argtypes = [vp, C.c_int, C.c_int]
    lib.pk_profile_sampling.argtypes = [vp, C.c_int]
    lib.pk_set_csr_map.argtypes = [vp, C.c_int, c_int32_p, c_int32_p, C.c_int64, C.c_int64]
    lib.pk_gather_csr_dev.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.pk_eval_jac_csr_dev.argtypes = [vp, vp, vp, vp]
    lib.pk_eval_hess_csr_dev.argtypes = [vp, vp, vp, C.c_double, vp, vp]
    lib.pk_eval_jac_csr.argtypes = [vp, dp, dp]
    lib.pk_eval_hess_csr.argtypes = [vp, dp, dp, C.c_double, dp]
    lib.pk_trace_read.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int64]
    lib.pk_eval_f_dev.argtypes = [vp, vp, vp, vp]
    lib.pk_eval_grad_dev.argtypes = [vp, vp, vp, vp]
    lib.pk_eval_g_dev.argtypes = [vp, vp, vp, vp]
    lib.pk_eval_jac_dev.argtypes = [vp, vp, vp, vp]
    lib.pk_eval_hess_dev.argtypes = [vp, vp, vp, C.c_double, vp, vp]
    lib.pk_eval_cycle_dev.argtypes = [vp, vp, vp, C.c_double, vp, vp, vp, vp, vp, vp]
    lib.pk_eval_hessc_prepared.argtypes = [vp, dp, C.c_double, dp, C.c_int]
    lib.pk_eval_cycle_dev_repeat.argtypes = [vp, vp, vp, C.c_double, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.pk_sync.argtypes = [vp, vp]
    lib.pk_wait_idle.argtypes = [vp, vp]
    lib.pk_same_bits.argtypes = [vp, vp, C.c_size_t]
    lib.pk_copy_bits.argtypes = [vp, vp, C.c_size_t]
    lib.pk_host_threads.argtypes = [C.c_int]
    lib.pk_host_threads_jobs.restype = C.c_long
    lib.pk_set_shard.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.pk_eval_integrals_dev.argtypes = [vp, vp, vp]
    lib.pk_eval_f_from_integrals_dev.argtypes = [vp, vp, vp, vp]
    lib.pk_aux_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.pk_eval_outer_dev.argtypes = [vp, vp, vp, vp]
    lib.pk_store_word_dev.argtypes = [vp, vp, C.c_int64, vp]
    lib.pk_profile.argtypes = [vp, C.c_int]
    lib.pk_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_int64), dp]
    lib.pk_kernel_name.argtypes = [C.c_int]
    lib.pk_kernel_name.restype = C.c_char_p
    for name in EXPORTS:
        if name not in ("pk_destroy", "pk_last_error", "pk_kernel_name"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def as_dp(a):
    return a.ctypes.data_as(c_double_p)


class Context:
    """One GPU context (pk_ctx)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.handle = C.c_void_p()
        rc = self.lib.pk_create(C.byref(self.handle), int(device))
        if rc != 0:
            msg = self.lib.pk_last_error(None).decode()
            self.handle = None
            raise RuntimeError(f"pk_create failed ({rc}): {msg}")

    def check(self, rc):
        if rc != 0:
            raise RuntimeError(f"libpockit_hip error {rc}: {self.lib.pk_last_error(self.handle).decode()}")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pk_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _PinnedBlock:
    """Owner of one pk_host_alloc allocation (freed with the last array over it)."""

    def __init__(self, nbytes):
        lib = load_library()
        self._lib, self.ptr = lib, C.c_void_p()
        rc = lib.pk_host_alloc(int(nbytes), C.byref(self.ptr))
        if rc != 0:
            raise RuntimeError(f"pk_host_alloc failed ({rc}): {lib.pk_last_error(None).decode()}")

    def buffer(self, count):
        buf = (C.c_double * count).from_address(self.ptr.value)
        buf._owner = self                        # every NumPy view keeps the ctypes array, which keeps the block
        return buf

    def __del__(self):
        try:
            if self.ptr:
                self._lib.pk_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class PinnedArray:
    """A float64 NumPy array over pinned (page-locked, device-visible) host memory: a DMA target at full PCIe rate that
    can be handed to the solver as a callback's result.  The memory is not tied to a context; it is released with the
    last view.  ``free()`` tells whether nobody but this object refers to the array any more (the caller dropped the
    result it was given, and every view of it), i.e. whether the next iterate may land in it."""

    def __init__(self, count, buffer_factory=None):
        import sys

        self._getrefcount = sys.getrefcount
        count = int(count)
        if buffer_factory is None:
            root = np.frombuffer(_PinnedBlock(8 * max(count, 1)).buffer(max(count, 1)), dtype=np.float64)
        else:
            root = buffer_factory(max(count, 1))   # (tests: plain NumPy memory)
        self.root = root
        self.array = root[:count]
        self.address = int(root.ctypes.data)
        self.ready = False       # (evaluator: the x-independent entries have been filled in)

    def free(self):
        # root: self.root + self.array.base (+ the argument of getrefcount); array: self.array (+ argument).  Views a
        # caller derived from the result keep ``root`` (NumPy collapses view chains to the memory's owner).
        return self._getrefcount(self.array) == 2 and self._getrefcount(self.root) == 3


class PinnedRing:
    """Result arrays of one output, recycled: ``take()`` returns an array nobody refers to any more, allocates a new
    one while fewer than ``cap`` exist, and returns None beyond that (the caller then falls back to a plain array and
    a host copy) -- a solver that keeps every iterate's Jacobian must not pin unbounded memory."""

    def __init__(self, count, cap=6, buffer_factory=None):
        self.count, self.cap, self._factory, self.items = int(count), int(cap), buffer_factory, []

    def take_item(self):
        """The ``PinnedArray`` itself (its ``array``, ``address``, ``ready``), or None when all ``cap`` are in use."""
        for it in self.items:
            if it.free():
                return it
        if len(self.items) < self.cap:
            self.items.append(PinnedArray(self.count, self._factory))
            return self.items[-1]
        return None

    def take(self):
        it = self.take_item()
        return None if it is None else it.array


# ---------------------------------------------------------------- helper threads for the host's passes over x / lambda
_HOST_HELPERS = {"k": None}       # None: not decided in this process yet; 0: tried and dropped; k: running


def host_helpers(lib, n, world=1, force=False):
    """Start the library's helper threads (``pk_host_threads``) for a solver thread whose x has ``n`` doubles -- once per process,
    only from 2 MB on, as many as this process's share of the host's cores allows (at most 6; POCKIT_AMD_HOST_THREADS=k
    overrides, 0 = none), and only if a measured pass over n doubles is at least a quarter faster with them.  Returns the
    number of helpers running.  A bitwise compare of x per callback and the staging copies of x and lambda are the host's
    share of an iterate: 90 us per pass at the 40k-node configuration."""
    import os
    import time

    env = os.environ.get("POCKIT_AMD_HOST_THREADS", "auto")
    # (``force``, or an explicit POCKIT_AMD_HOST_THREADS=k: also for an x below 2 MB -- the library uses helpers from 256 KB per
    #  pass on.  Not the default there: the helpers spin for a millisecond after every pass, which a tight loop of small
    #  iterates turns into permanently busy cores)
    if env == "0" or (8 * n < (2 << 20) and not force and not env.isdigit()):
        return _HOST_HELPERS["k"] or 0
    if _HOST_HELPERS["k"] is not None:
        return _HOST_HELPERS["k"]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    k = int(env) if env.isdigit() else max(0, min(6, cores // max(world, 1) - 2))
    _HOST_HELPERS["k"] = 0
    if k < 1:
        return 0
    a = np.zeros(n)
    b = np.zeros(n)
    pa, pb = a.ctypes.data, b.ctypes.data

    def pass_us():
        lib.pk_same_bits(pa, pb, n)
        ts = []
        for _ in range(9):
            t = time.perf_counter()
            lib.pk_same_bits(pa, pb, n)
            ts.append(time.perf_counter() - t)
        return sorted(ts)[len(ts) // 2]

    lib.pk_host_threads(0)
    alone = pass_us()
    if lib.pk_host_threads(k):
        return 0
    helped = pass_us()
    if helped > 0.75 * alone:
        lib.pk_host_threads(0)
        return 0
    _HOST_HELPERS["k"] = k
    return k


def host_helpers_stopped():
    """(somebody called pk_host_threads(0): the next large evaluator may try again)"""
    _HOST_HELPERS["k"] = None
