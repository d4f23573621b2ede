"""Bindings of the native code.

`ops`  — the product path: torch.ops.magnet.* of libmagnet_torch.so, the thin PyTorch-ROCm C++
         extension (csrc/torch_ext.cpp) that takes tensors, checks them, and forwards raw device
         pointers + the current torch stream to the C ABI of libmagnet_hip.so (include/magnet_hip.h).
`lib`  — the same C ABI bound directly with ctypes: what a caller without torch would bind, used
         here for the ABI checks and by the tests that call entry points with raw pointers.

Both libraries are built in-tree; if either is missing the import fails loudly.  There is no CPU
fallback: the operators are registered for the GPU dispatch key only.
"""
import ctypes as C
import os

import torch

class MagnetHipError(RuntimeError):
    pass


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libmagnet_hip.so')
EXT_PATH = os.path.join(_HERE, 'csrc', 'libmagnet_torch.so')


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libmagnet_hip.so is not built (%s). Run `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C magnet_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    i32, i64, u64, f32, f64, vp, sz = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_void_p, C.c_size_t
    lib.mh_version.restype = C.c_int
    lib.mh_last_error.restype = C.c_char_p
    lib.mh_lap_workspace_bytes.restype = sz
    lib.mh_lap_workspace_bytes.argtypes = [i64, i32, C.c_int, C.c_int]
    lib.mh_lap_build_count.restype = C.c_int
    lib.mh_lap_build_count.argtypes = [vp, vp, C.c_int, vp, i64, i32, f64, C.c_int, vp, sz, vp,
                                       C.POINTER(i64), vp]
    lib.mh_lap_build_fill.restype = C.c_int
    lib.mh_lap_build_fill.argtypes = [i64, i32, f64, f64, C.c_int, C.c_int, vp, sz, i64, vp, vp, vp, vp]
    lib.mh_lap_block_workspace_bytes.restype = sz
    lib.mh_lap_block_workspace_bytes.argtypes = [i64, i64, i32, C.c_int]
    lib.mh_lap_block_count.restype = C.c_int
    lib.mh_lap_block_count.argtypes = [vp, vp, vp, i64, vp, vp, vp, i64, C.c_int, C.c_int, i32, i32, i32, f64, vp, sz, vp, vp,
                                       C.POINTER(i64), vp]
    lib.mh_lap_block_count_tagged.restype = C.c_int
    lib.mh_lap_block_count_tagged.argtypes = [vp, vp, vp, i64, vp, vp, vp, i64, C.c_int, C.c_int, i32, i32, i32, f64, vp, vp, sz, vp,
                                              vp, C.POINTER(i64), vp]
    lib.mh_lap_block_fill.restype = C.c_int
    lib.mh_lap_block_fill.argtypes = [i64, i64, i32, i32, i32, f64, f64, C.c_int, vp, sz, i64, vp, vp, vp, vp]
    lib.mh_cspmm.restype = C.c_int
    lib.mh_cspmm.argtypes = [vp, vp, vp, i32,
                             vp, i64, i64,
                             vp, i64, i64, f32,
                             vp, i64, i64, f32,
                             vp, i64, i64, f32,
                             vp, C.c_int, vp, i32, vp]
    lib.mh_cspmm_planned.restype = C.c_int
    lib.mh_cspmm_planned.argtypes = lib.mh_cspmm.argtypes[:-1] + [vp, i32, i32, i32, vp, sz, vp]
    lib.mh_cspmm_planned_noalloc.restype = C.c_int
    lib.mh_cspmm_planned_noalloc.argtypes = lib.mh_cspmm.argtypes[:-1] + [vp, i32, i32, i32, vp]
    lib.mh_cspmm_plan_ints.restype = sz
    lib.mh_cspmm_plan_ints.argtypes = [i32, i32]
    lib.mh_cspmm_plan_count.restype = C.c_int
    lib.mh_cspmm_plan_count.argtypes = [vp, i32, i32, vp, C.POINTER(i32), C.POINTER(i32), vp]
    lib.mh_cspmm_plan_fill.restype = C.c_int
    lib.mh_cspmm_plan_fill.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.mh_cheb_terms.restype = C.c_int
    lib.mh_cheb_terms.argtypes = [vp, vp, vp, i32, vp, i64, i32, i32, vp, i32, i32, i32, vp, sz, vp]
    lib.mh_clenshaw.restype = C.c_int
    lib.mh_clenshaw.argtypes = [vp, vp, vp, i32, vp, i64, i32, vp, vp, vp, C.c_int, vp, i32, vp, i32, i32, i32, vp, sz, vp]
    lib.mh_unpack.restype = C.c_int
    lib.mh_unpack.argtypes = [vp, i64, i64, vp, i64, vp, i64, i64, i32, vp]
    lib.mh_gather_rows.restype = C.c_int
    lib.mh_gather_rows.argtypes = [vp, i64, i64, vp, i64, vp, i32, vp]
    lib.mh_scatter_rows.restype = C.c_int
    lib.mh_scatter_rows.argtypes = [vp, vp, i64, vp, i64, i64, i32, vp]
    lib.mh_halo_plan_workspace_bytes.restype = sz
    lib.mh_halo_plan_workspace_bytes.argtypes = [i32]
    lib.mh_halo_plan_count.restype = C.c_int
    lib.mh_halo_plan_count.argtypes = [vp, i64, i32, i32, i32, vp, sz, C.POINTER(i64), vp]
    lib.mh_halo_plan_fill.restype = C.c_int
    lib.mh_halo_plan_fill.argtypes = [vp, i64, i32, i32, i32, vp, i32, vp, sz, i64, vp, vp, vp, vp]
    lib.mh_pack.restype = C.c_int
    lib.mh_pack.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i32, vp]
    lib.mh_bias_relu.restype = C.c_int
    lib.mh_bias_relu.argtypes = [vp, i64, i64, vp, C.c_int, vp, i64, i32, vp]
    lib.mh_relu_bwd.restype = C.c_int
    lib.mh_relu_bwd.argtypes = [vp, i64, i64, vp, i64, i32, vp]
    lib.mh_relu_bwd_to.restype = C.c_int
    lib.mh_relu_bwd_to.argtypes = [vp, i64, i64, vp, vp, i64, i64, i64, i32, vp]
    lib.mh_cspmm_noalloc_init.restype = C.c_int
    lib.mh_cspmm_noalloc_init.argtypes = []
    lib.mh_cspmm_set_grid_cap.restype = C.c_int
    lib.mh_cspmm_set_grid_cap.argtypes = [i32]
    lib.mh_exchange_id_bytes.restype = C.c_int
    lib.mh_exchange_id_bytes.argtypes = []
    lib.mh_exchange_new_id.restype = C.c_int
    lib.mh_exchange_new_id.argtypes = [vp, sz]
    lib.mh_exchange_create.restype = C.c_int
    lib.mh_exchange_create.argtypes = [vp, sz, i32, i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), i64, C.POINTER(vp)]
    lib.mh_exchange_begin.restype = C.c_int
    lib.mh_exchange_begin.argtypes = [vp, vp, i64, i64, vp, i32, vp]
    lib.mh_exchange_end.restype = C.c_int
    lib.mh_exchange_end.argtypes = [vp, vp]
    lib.mh_exchange_all_gather.restype = C.c_int
    lib.mh_exchange_all_gather.argtypes = [vp, vp, vp, i64, vp]
    lib.mh_exchange_destroy.restype = C.c_int
    lib.mh_exchange_destroy.argtypes = [vp]
    lib.mh_powerlaw_edges.restype = C.c_int
    lib.mh_powerlaw_edges.argtypes = [i32, i64, u64, i64, vp, vp, vp]
    lib.mh_stream_read.restype = C.c_int
    lib.mh_stream_read.argtypes = [vp, sz, vp, vp]
    return lib


lib = _load()

# operators the extension registers (csrc/torch_ext.cpp)
OPS = ['cspmm', 'cspmm_noalloc', 'plan_build', 'cheb_terms_', 'clenshaw', 'lap_build', 'lap_block_count',
       'lap_block_fill', 'pack', 'unpack',
       'gather_rows', 'scatter_rows', 'bias_relu_', 'relu_bwd_', 'relu_bwd_out', 'halo_plan', 'powerlaw_edges', 'stream_read',
       'set_spmm_grid_cap', 'noalloc_init', 'exchange_new_id', 'exchange_create', 'exchange_begin', 'exchange_end',
       'exchange_all_gather', 'exchange_destroy', 'version']


def _guarded(op):
    """The operator with its errors raised as MagnetHipError (a RuntimeError, as the extension throws them)."""
    def call(*args):
        try:
            return op(*args)
        except NotImplementedError as e:            # no kernel for that backend: CPU tensors
            raise MagnetHipError('magnet_amd runs on the GPU only: %s' % str(e).split('\n')[0]) from None
        except RuntimeError as e:
            raise MagnetHipError(str(e).split('\nException raised from')[0]) from None
    call.__name__ = getattr(op, '__name__', 'op')
    return call


class _Ops:
    def __init__(self):
        if not os.path.exists(EXT_PATH):
            raise ImportError(
                'libmagnet_torch.so is not built (%s). Run `python -c "import __graft_entry__ as g; g.build()"` '
                'or `make -C magnet_amd/csrc`. There is no CPU fallback.' % EXT_PATH)
        torch.ops.load_library(EXT_PATH)
        ns = torch.ops.magnet
        for name in OPS:
            setattr(self, name, _guarded(getattr(ns, name).default))


ops = _Ops()

# every symbol include/magnet_hip.h declares (tests/test_abi_cpu.py compares the two lists)
EXPORTS = [
    'mh_version', 'mh_last_error',
    'mh_lap_workspace_bytes', 'mh_lap_build_count', 'mh_lap_build_fill',
    'mh_lap_block_workspace_bytes', 'mh_lap_block_count', 'mh_lap_block_count_tagged', 'mh_lap_block_fill',
    'mh_cspmm', 'mh_cspmm_planned', 'mh_cspmm_planned_noalloc', 'mh_cspmm_noalloc_init', 'mh_cspmm_set_grid_cap', 'mh_cspmm_plan_ints', 'mh_cspmm_plan_count', 'mh_cspmm_plan_fill',
    'mh_cheb_terms', 'mh_clenshaw',
    'mh_halo_plan_workspace_bytes', 'mh_halo_plan_count', 'mh_halo_plan_fill',
    'mh_pack', 'mh_unpack', 'mh_gather_rows', 'mh_scatter_rows', 'mh_bias_relu', 'mh_relu_bwd', 'mh_relu_bwd_to',
    'mh_exchange_id_bytes', 'mh_exchange_new_id', 'mh_exchange_create', 'mh_exchange_begin', 'mh_exchange_end',
    'mh_exchange_all_gather', 'mh_exchange_destroy',
    'mh_powerlaw_edges', 'mh_stream_read',
]


def check(rc):
    if rc != 0:
        raise MagnetHipError('libmagnet_hip error %d: %s' % (rc, lib.mh_last_error().decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MagnetHipError('magnet_amd runs on the GPU only: got a %s tensor' % t.device)
