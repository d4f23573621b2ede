"""CPU-side checks of the C ABI: the library loads, exports every symbol the header declares,
and rejects bad arguments before touching a device.  No compute is launched."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, 'include', 'magnet_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mh_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from magnet_amd import _lib
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(_lib.lib, n), 'libmagnet_hip.so does not export %s' % n
    assert sorted(_lib.EXPORTS) == names
    assert _lib.lib.mh_version() >= 200


def test_torch_extension_registers_every_operator():
    """The product path is torch.ops.magnet.* of libmagnet_torch.so; it has kernels for the GPU dispatch key
    only, so CPU tensors are refused instead of being computed somewhere else."""
    import torch
    from magnet_amd import _lib, MagnetHipError
    assert os.path.exists(_lib.EXT_PATH)
    for name in _lib.OPS:
        assert hasattr(torch.ops.magnet, name), 'torch.ops.magnet.%s is not registered' % name
    assert _lib.ops.version() == _lib.lib.mh_version() >= 200
    src = open(os.path.join(ROOT, 'magnet_amd', 'csrc', 'torch_ext.cpp')).read()
    assert sorted(set(re.findall(r'm\.def\("([a-z_]+)\(', src))) == sorted(_lib.OPS)
    with pytest.raises(MagnetHipError):
        _lib.ops.bias_relu_(torch.zeros(3, 2, 4), None, True, None)
    with pytest.raises(MagnetHipError):
        _lib.ops.lap_build(torch.zeros(3, dtype=torch.int64), torch.zeros(3, dtype=torch.int64), None, 4, 0.25, 2.0, False)


def test_bad_arguments_are_rejected_without_a_device():
    from magnet_amd import _lib
    lib = _lib.lib
    rc = lib.mh_cspmm(None, None, None, -1, None, 0, 0, None, 0, 0, 1.0, None, 0, 0, 0.0, None, 0, 0, 0.0,
                      None, 0, None, 4, None)
    assert rc == -1 and b'n_rows' in lib.mh_last_error()
    rc = lib.mh_cspmm(None, None, None, 5, None, 4, 4, None, 4, 4, 1.0, None, 0, 0, 0.0, None, 0, 0, 0.0,
                      None, 0, None, 4, None)
    assert rc == -1 and b'null' in lib.mh_last_error()
    nnz = C.c_int64(0)
    rc = lib.mh_lap_build_count(None, None, 1, None, 10, 5, 2.0, 0, None, 0, None, C.byref(nnz), None)
    assert rc == -1
    assert lib.mh_pack(None, 0, None, 0, None, 0, 0, -3, 4, None) == -1
    assert lib.mh_powerlaw_edges(0, 10, 1, 0, None, None, None) == -1
    with pytest.raises(_lib.MagnetHipError):
        _lib.check(rc)


def test_workspace_size_grows_with_the_graph():
    from magnet_amd import _lib
    a = _lib.lib.mh_lap_workspace_bytes(1000, 100, 0, 0)
    b = _lib.lib.mh_lap_workspace_bytes(100000, 100, 0, 0)
    c = _lib.lib.mh_lap_workspace_bytes(100000, 100, 1, 0)
    assert 0 < a < b < c


def test_product_path_refuses_cpu_tensors():
    import torch
    from magnet_amd import MagnetHipError, pack_complex
    with pytest.raises(MagnetHipError):
        pack_complex(torch.zeros(3, 4), torch.zeros(3, 4))


def test_int32_range_is_checked_before_any_memory_is_touched():
    from magnet_amd import _lib
    lib = _lib.lib
    fake = C.c_void_p(4096)
    nnz = C.c_int64(0)
    rc = lib.mh_lap_build_count(fake, fake, 0, None, 2 ** 30 + 5, 1000, 2.0, 0, fake, 1 << 62, fake, C.byref(nnz), None)
    assert rc == -5 and b'int32' in lib.mh_last_error()
    rc = lib.mh_lap_build_count(fake, fake, 0, None, 1000, 1000, 2.0, 0, fake, 16, fake, C.byref(nnz), None)
    assert rc == -4 and b'workspace' in lib.mh_last_error()
    rc = lib.mh_cspmm_planned(fake, fake, fake, 10, fake, 8, 4, C.c_void_p(8192), 8, 4, 1.0, None, 0, 0, 0.0, None, 0, 0, 0.0,
                              None, 0, None, 4, fake, 3, 2, 512, fake, 1 << 20, None)
    assert rc == -1 and b'plan' in lib.mh_last_error()


def test_header_is_plain_c():
    """include/magnet_hip.h is what a C caller binds: it must compile as C99 on its own."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, 'use.c')
        open(src, 'w').write('#include "magnet_hip.h"\nint main(void) { return mh_version() > 0 ? 0 : 1; }\n')
        res = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only',
                              '-I', os.path.join(ROOT, 'include'), src], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_exchange_entry_points_check_their_arguments_and_report_a_missing_rccl():
    """mh_exchange_*: bad plans are refused before anything is loaded; a communication library that cannot be loaded is
    an error code with a message (MH_ERR_COMM), not a crash, and leaves the rest of the library usable."""
    import subprocess
    import sys
    from magnet_amd import _lib
    lib = _lib.lib
    assert lib.mh_exchange_id_bytes() == 128
    small = (C.c_ubyte * 16)()
    assert lib.mh_exchange_new_id(small, 16) == -1 and b'128' in lib.mh_last_error()
    ident = (C.c_ubyte * 128)()
    out = C.c_void_p()
    rows = (C.c_void_p * 2)(None, None)
    count = (C.c_int64 * 2)(0, 0)
    ptr = (C.c_int64 * 3)(0, 0, 0)
    assert lib.mh_exchange_create(ident, 128, 2, 2, rows, count, ptr, 10, C.byref(out)) == -1      # rank outside the world
    count[1] = 11
    assert lib.mh_exchange_create(ident, 128, 0, 2, rows, count, ptr, 10, C.byref(out)) == -1      # more rows than the block has
    assert b'give_count' in lib.mh_last_error() and not out.value
    assert lib.mh_exchange_begin(None, None, 0, 0, None, 4, None) == -1
    assert lib.mh_exchange_end(None, None) == -1
    assert lib.mh_exchange_destroy(None) == 0
    code = ("import ctypes as C\nfrom magnet_amd import _lib\nb = (C.c_ubyte * 128)()\n"
            "rc = _lib.lib.mh_exchange_new_id(b, 128)\nprint(rc, _lib.lib.mh_last_error().decode())\n"
            "assert _lib.lib.mh_version() >= 300\n")
    env = dict(os.environ, MAGNET_RCCL_LIBRARY='/nonexistent/librccl.so')
    res = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-500:]
    assert res.stdout.startswith('-6 ') and 'could not be loaded' in res.stdout, res.stdout
