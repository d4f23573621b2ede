#!/bin/bash
# GPU box (one GPU): bench.py as the driver launches it for --gpus 8, all eight ranks on this GPU over gloo (transfers
# staged through host memory; the library-posted exchange over the shared-memory stand-in for RCCL).  A rehearsal of the
# eight-rank control flow and bookkeeping; its times mean nothing.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/run
mkdir -p $O
S=${1:-auto}
export MH_DIST_BACKEND=gloo MAGNET_RCCL_LIBRARY=$PWD/tests/abi/libmock_rccl.so MOCK_RCCL_SLOT_BYTES=33554432 OMP_NUM_THREADS=2
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29641 \
  bench.py --gpus 8 --steps 2 --warmup 1 --nodes 200000 --edges 4000000 --features 64 --K 3 --strategy $S \
  --cpu-sample-nodes 20000 --cpu-sample-edges 200000 --epoch-nodes 50000 --epoch-edges 1000000 > $O/reh8_$S.json 2> $O/reh8_$S.err
echo "exit $?"
python3 - $S <<'PY'
import json, sys
d = [json.loads(l) for l in open('gpurun_out/run/reh8_%s.json' % sys.argv[1]) if l.startswith('{')][0]
m = d['multi_gpu']
print(d['config']['parallelism'], 'value %.3g' % d['value'], 'ranks', m['ranks_initialised'])
print({k: (round(v['ms_per_step'], 1) if 'ms_per_step' in v else v) for k, v in m['row_variants'].items() if isinstance(v, dict)})
print('epoch', {k: (v.get('ms'), v.get('error')) for k, v in d['epoch']['layouts'].items()}, 'cpu', d['cpu_baseline'].get('value'))
PY
tail -3 $O/reh8_$S.err
