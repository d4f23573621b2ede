"""mh_cspmm_planned_noalloc (hub-row partial sums in the library's built-in per-GPU buffer) against
mh_cspmm_planned (partial sums in caller memory), both called through the C ABI: bit-equal outputs on
two calls in a row, the second on a smaller input, and a clear refusal beyond the built-in size."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(n, e, f, seed, dev):
    import torch
    from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges
    src, dst = powerlaw_edges(n, e, seed, device=dev)
    lap = build_magnetic_laplacian(src, dst, n, q=0.25)
    plan, n_long, n_chunks, chunk = lap.plan()
    assert n_long > 0 and n_chunks > n_long, 'the case must have hub rows (got n_long=%d)' % n_long
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(n, 2, f, device=dev, generator=g)
    z = torch.randn(n, 2, f, device=dev, generator=g)
    bias = torch.randn(f, device=dev, generator=g)
    return lap, plan, n_long, n_chunks, chunk, x, z, bias


def _run(entry, lap, plan, n_long, n_chunks, chunk, x, z, bias, scratch):
    import torch
    from magnet_amd import _lib
    from magnet_amd._lib import ptr, stream_ptr
    n, _, f = x.shape
    y = torch.full((n, 2, f), float('nan'), device=x.device)
    mask = torch.zeros((n, f), dtype=torch.uint8, device=x.device)
    args = [ptr(lap.row_ptr), ptr(lap.col_idx), ptr(lap.val), n, ptr(x), 2 * f, f, ptr(y), 2 * f, f, 2.0,
            ptr(z), 2 * f, f, -1.0, None, 0, 0, 0.0, ptr(bias), 1, ptr(mask), f, ptr(plan), n_long, n_chunks, chunk]
    if scratch is not None:
        args += [ptr(scratch), scratch.numel() * 4]
    rc = getattr(_lib.lib, entry)(*args, stream_ptr(x.device))
    torch.cuda.synchronize()
    return rc, y, mask


def test_noalloc_entry_point_matches_planned_twice():
    import torch
    from magnet_amd import _lib
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    # first call: the larger input; second call: a smaller one (fewer chunks, narrower rows), so stale
    # partial sums of the first call sit right behind the slots the second call uses
    for (n, e, f, seed) in ((60_000, 1_500_000, 64, 7), (20_000, 300_000, 32, 8)):
        lap, plan, n_long, n_chunks, chunk, x, z, bias = _case(n, e, f, seed, dev)
        scratch = torch.empty(n_chunks * 2 * f, dtype=torch.float32, device=dev)
        rc0, y0, m0 = _run('mh_cspmm_planned', lap, plan, n_long, n_chunks, chunk, x, z, bias, scratch)
        rc1, y1, m1 = _run('mh_cspmm_planned_noalloc', lap, plan, n_long, n_chunks, chunk, x, z, bias, None)
        assert rc0 == 0 and rc1 == 0, _lib.lib.mh_last_error()
        assert not torch.isnan(y0).any()
        assert torch.equal(y0, y1), 'outputs differ (n=%d)' % n
        assert torch.equal(m0, m1)
        # and against a float64 product on the host, on a row sample that includes the hub rows
        Ls = lap.to_scipy()
        rows = np.unique(np.concatenate([np.arange(0, 64), np.random.default_rng(seed).integers(0, n, 256)]))
        xc = (x[:, 0] + 1j * x[:, 1]).cpu().numpy().astype(np.complex128)
        zc = (z[:, 0] + 1j * z[:, 1]).cpu().numpy().astype(np.complex128)
        ref = 2.0 * (Ls[rows].astype(np.complex128) @ xc) - zc[rows] + bias.cpu().numpy().astype(np.float64) * (1 + 1j)
        gate = ref.real >= 0
        ref = ref * gate
        got = (y1[:, 0] + 1j * y1[:, 1]).cpu().numpy()[rows]
        far = np.abs(ref.real) > 1e-4                       # entries whose gate an fp32 rounding cannot flip
        assert np.abs(got - ref)[far].max() <= 1e-5 * np.abs(ref).max()


def test_noalloc_entry_point_under_hip_graph_capture():
    """What the entry point is for: a caller that may not allocate.  After one ordinary call on the device (which
    resolves the built-in buffer's address), the operator is captured into a HIP graph and replayed on new inputs;
    every replay equals the planned entry point run eagerly, and the capture itself allocates no device memory."""
    import torch
    from magnet_amd._lib import ops
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    lap, plan, n_long, n_chunks, chunk, x, z, bias = _case(40_000, 900_000, 64, 9, dev)
    n, _, f = x.shape
    y = torch.empty_like(x); mask = torch.empty((n, f), dtype=torch.uint8, device=dev)
    call = lambda: ops.cspmm_noalloc(lap.row_ptr, lap.col_idx, lap.val, n, x, y, 2.0, z, -1.0, None, 0.0, bias, True, mask,
                                     plan, n_long, n_chunks, chunk)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call()                                    # first call on this device: outside any capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    def captured_bytes(body):                     # what a capture adds to torch's allocator (its own bookkeeping included)
        graph = torch.cuda.CUDAGraph()
        before = torch.cuda.memory_allocated(dev)
        with torch.cuda.graph(graph):             # global capture mode: a hipMalloc in here would fail the capture
            body()
        return graph, torch.cuda.memory_allocated(dev) - before
    idle, _ = captured_bytes(lambda: y.add_(0.0))     # holds the generator state torch sets up with a process's first graph
    graph, used = captured_bytes(call)
    assert used == 0
    g = torch.Generator(device=dev).manual_seed(77)
    for _ in range(3):
        x.copy_(torch.randn(x.shape, device=dev, generator=g))
        y.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        want = lap.spmm(x, alpha=2.0, z1=z, beta1=-1.0, bias=bias, relu=True, mask=torch.empty_like(mask))
        assert torch.equal(y, want)


def test_noalloc_entry_point_refuses_what_does_not_fit():
    import torch
    from magnet_amd import _lib
    dev = torch.device('cuda', 0)
    fake = C.c_void_p(1 << 20)
    n_chunks = (64 << 20) // (2 * 256 * 4) + 1
    rc = _lib.lib.mh_cspmm_planned_noalloc(fake, fake, fake, 10, fake, 512, 256, C.c_void_p(2 << 20), 512, 256, 1.0,
                                           None, 0, 0, 0.0, None, 0, 0, 0.0, None, 0, None, 256, fake, 1, n_chunks, 512,
                                           _lib.stream_ptr(dev))
    assert rc == -4 and b'built-in scratch' in _lib.lib.mh_last_error()
