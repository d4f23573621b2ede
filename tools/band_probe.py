#!/usr/bin/env python3
"""Why the config-5 SpMM lands at 146 ms per order in some processes and at 158-162 ms in others.
One process records, next to the time of every order: which GPU it ran on (uuid), the memory / fabric / shader
clocks and the socket power sampled from sysfs WHILE the kernels run, the streaming-read ceiling
(mh_stream_read) measured in the same process, the placement (address modulo 2 MiB) of the feature block and of
the CSR arrays, and the same orders again after the block has been freed and re-allocated behind a pad."""
import glob
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnet_amd import _lib, laplacian as lapmod                                # noqa: E402
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges       # noqa: E402
from magnet_amd.layer.sparse_magnet import _fill_orders                         # noqa: E402

dev = torch.device('cuda', 0)
props = torch.cuda.get_device_properties(0)


def sysfs_card():
    for d in sorted(glob.glob('/sys/class/drm/card*/device')):
        if os.path.exists(d + '/pp_dpm_mclk'):
            return d
    return None


CARD = sysfs_card()


def current_level(path):
    try:
        for line in open(path):
            if '*' in line:
                return float(line.split(':')[1].strip().lower().replace('mhz', '').replace('*', '').strip())
    except Exception:
        pass
    return None


def read_power():
    for p in glob.glob((CARD or '') + '/hwmon/hwmon*/power1_*'):
        if p.endswith(('power1_average', 'power1_input')):
            try:
                return float(open(p).read()) / 1e6
            except Exception:
                pass
    return None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop and CARD:
            self.rows.append((current_level(CARD + '/pp_dpm_sclk'), current_level(CARD + '/pp_dpm_mclk'),
                              current_level(CARD + '/pp_dpm_fclk'), read_power()))
            time.sleep(0.02)

    def summary(self):
        out = {}
        for i, name in enumerate(('sclk_mhz', 'mclk_mhz', 'fclk_mhz', 'power_w')):
            v = [r[i] for r in self.rows if r[i] is not None]
            out[name] = None if not v else {'min': min(v), 'mean': sum(v) / len(v), 'max': max(v), 'samples': len(v)}
        return out


n, e, f, K = 10_000_000, 200_000_000, 256, 3
src, dst = powerlaw_edges(n, e, 43, device=dev)
lap = build_magnetic_laplacian(src, dst, n, q=0.25)
del src, dst
torch.cuda.empty_cache()


def read_ceiling(nbytes=8 << 30):
    a = torch.zeros(nbytes // 4, device=dev); sink = torch.zeros(1, device=dev)
    _lib.ops.stream_read(a, sink)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _lib.ops.stream_read(a, sink)
    e1.record(); torch.cuda.synchronize()
    return nbytes * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def orders(blk, reps=6):
    launches = []
    lapmod.SPMM_OBSERVER = lambda a, b, info: launches.append((a, b))
    _fill_orders(lap, blk); torch.cuda.synchronize(); launches.clear()
    s = Sampler(); s.start()
    for _ in range(reps):
        _fill_orders(lap, blk)
    torch.cuda.synchronize()
    s.stop = True; s.join()
    lapmod.SPMM_OBSERVER = None
    ms = [a.elapsed_time(b) for a, b in launches]
    return {'ms_per_order_mean': sum(ms) / len(ms), 'ms_per_order_min': min(ms), 'ms_per_order_max': max(ms), 'clocks': s.summary()}


mod = lambda t: t.data_ptr() % (2 << 20)
out = {'gpu': props.name, 'uuid': str(getattr(props, 'uuid', 'n/a')), 'sysfs_card': CARD}
blk = torch.empty(n, 2, K + 1, f, device=dev); blk[:, :, 0, :].normal_()
out['placement'] = {'block_mod_2MiB': mod(blk), 'col_idx_mod_2MiB': mod(lap.col_idx), 'val_mod_2MiB': mod(lap.val), 'row_ptr_mod_2MiB': mod(lap.row_ptr)}
out['read_ceiling_GBs_before'] = read_ceiling()
out['first_allocation'] = orders(blk)
del blk
torch.cuda.empty_cache()
pad = torch.empty((1 << 30) + 4096 * 37, dtype=torch.uint8, device=dev)          # shifts where the block lands
blk = torch.empty(n, 2, K + 1, f, device=dev); blk[:, :, 0, :].normal_()
out['placement_second'] = {'block_mod_2MiB': mod(blk)}
out['second_allocation'] = orders(blk)
out['read_ceiling_GBs_after'] = read_ceiling()
print(json.dumps(out))
