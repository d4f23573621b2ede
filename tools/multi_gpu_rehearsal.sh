#!/bin/bash
# GPU box: single-rank rehearsals of the multi-GPU bench path on the real RCCL backend, and a 2-rank gloo rehearsal at a larger size
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/run
for s in rows cols; do
  timeout -k 10 500 python3 bench.py --force-dist --strategy $s --no-cpu-baseline > gpurun_out/run/bench_fd_$s.json 2> gpurun_out/run/bench_fd_$s.err || { tail -8 gpurun_out/run/bench_fd_$s.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/run/bench_fd_$s.json'))
print('$s', 'value %.4g ms/step %.1f' % (d['value'], d['ms_per_step']), 'epoch', {k: (v.get('ms'), v.get('ms_with_cached_input_propagation')) for k, v in d['epoch'].get('layouts', {}).items()}, 'variants', {k: round(v['ms_per_step'], 1) for k, v in d['multi_gpu']['row_variants'].items() if isinstance(v, dict)})"
done
# the leg whose exchanges the library posts itself is carried by the shared-memory stand-in for RCCL (one GPU here)
MAGNET_RCCL_LIBRARY=$PWD/tests/abi/libmock_rccl.so MOCK_RCCL_SLOT_BYTES=1610612736 MH_DIST_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 2 --warmup 1 --nodes 2000000 --edges 40000000 --features 128 --K 2 > gpurun_out/run/bench_gloo2.json 2> gpurun_out/run/bench_gloo2.err || { tail -8 gpurun_out/run/bench_gloo2.err; exit 1; }
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/run/bench_gloo2.json') if l.startswith('{')][-1])
print('gloo2', d['multi_gpu']['strategy'], 'value %.4g' % d['value'], {k: (round(v['ms_per_step'], 1), v['exposed_wait_ms_per_order']) for k, v in d['multi_gpu']['row_variants'].items() if isinstance(v, dict)}, 'epoch', {k: v.get('ms') for k, v in d['epoch']['layouts'].items()}, 'cpu', d['cpu_baseline'].get('value'))"
