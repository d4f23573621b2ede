#!/usr/bin/env python3
"""Does the weight GEMM of a conv hide under its last propagation orders when it runs on a few compute units only?
The step of bench.py (config 5: K = 3 SpMM orders into the side-by-side block, then one GEMM over the 4 orders) against a
variant that starts the GEMM over orders 0..1 on a stream restricted to CUS compute units (hipExtStreamCreateWithCUMask)
as soon as order 1 exists, and adds the GEMM over orders 2..3 at the end."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges

n, e, f, K = int(os.environ.get('N', 10_000_000)), int(os.environ.get('E', 200_000_000)), int(os.environ.get('F', 256)), 3
dev = torch.device('cuda', 0)
src, dst = powerlaw_edges(n, e, 43, device=dev)
lap = build_magnetic_laplacian(src, dst, n, q=0.25)
del src, dst
block = torch.empty(n, 2, K + 1, f, device=dev)
block[:, :, 0, :].normal_()
W = torch.randn((K + 1) * f, f, device=dev) / f ** 0.5
out = torch.empty(2 * n, f, device=dev)
rows = block.view(2 * n, (K + 1) * f)
order = lambda k: block[:, :, k, :]

def spmm(k):
    if k == 1:
        lap.spmm(order(0), out=order(1))
    else:
        lap.spmm(order(k - 1), out=order(k), alpha=2.0, z1=order(k - 2), beta1=-1.0)

def plain():
    for k in range(1, K + 1):
        spmm(k)
    torch.mm(rows, W, out=out)

# the HIP runtime already in the process (torch ships its own copy; a second copy would be a second runtime)
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
def masked_stream(cus, total=256):
    # the lowest `cus` bits of the mask (patterns with holes are refused by the driver on this box)
    words = (ctypes.c_uint32 * (total // 32))()
    for i in range(cus):
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(total // 32), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)

def overlapped(side, split):
    cur = torch.cuda.current_stream()
    for k in range(1, split + 1):
        spmm(k)
    ready = torch.cuda.Event(); ready.record(cur)
    with torch.cuda.stream(side):
        side.wait_event(ready)
        torch.mm(rows[:, :(split + 1) * f], W[:(split + 1) * f], out=out)
        done = torch.cuda.Event(); done.record(side)
    for k in range(split + 1, K + 1):
        spmm(k)
    cur.wait_event(done)
    out.addmm_(rows[:, (split + 1) * f:], W[(split + 1) * f:])

def timeit(fn, iters=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

print('plain step           %.1f ms' % timeit(plain), flush=True)
ref = out.clone()
for cus in (16, 32, 64, 128):
    side = masked_stream(cus)
    for split in (1, 2):
        t = timeit(lambda: overlapped(side, split))
        err = float((out - ref).abs().max() / ref.abs().max())
        print('GEMM of orders 0..%d on %3d CUs under the later orders: %.1f ms (max rel diff of the output %.1e)' % (split, cus, t, err), flush=True)
print('plain step again     %.1f ms' % timeit(plain), flush=True)
