"""bench.py pieces that run without a GPU."""
import subprocess
import sys

from conftest import ROOT


def test_algorithmic_bytes_formula():
    sys.path.insert(0, ROOT)
    import bench
    nnz, n, f = 400_000_000, 10_000_000, 256
    # CSR once (12 B per nonzero + row_ptr) + one gathered 2 KiB row per nonzero + one output row per node
    assert bench.algorithmic_bytes(nnz, n, f, 0) == nnz * 12 + 4 * (n + 1) + nnz * 2048 + n * 2048
    assert bench.algorithmic_bytes(nnz, n, f, 1) - bench.algorithmic_bytes(nnz, n, f, 0) == n * 2048
    assert bench.HBM_PEAK_GBS == 8000.0
    # compulsory bound: CSR once, X once, output once (+ one row per addend)
    assert bench.compulsory_bytes(nnz, n, f, 1) == nnz * 12 + 4 * (n + 1) + n * 2048 * 3
    assert bench.compulsory_bytes(nnz, n, f, 1) < bench.algorithmic_bytes(nnz, n, f, 1) / 10
    # builder model: 6 radix passes of 3 x 8 bytes per record dominate
    e = 200_000_000
    b = bench.builder_bytes(e, n, nnz, 6)
    assert 6 * 24 * 2 * e < b < 1.3 * 6 * 24 * 2 * e


def test_bench_help_runs_without_gpu():
    out = subprocess.run([sys.executable, 'bench.py', '--help'], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and '--gpus' in out.stdout and '--strategy' in out.stdout
