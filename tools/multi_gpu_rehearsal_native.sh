#!/bin/bash
# GPU box (one GPU): bench.py as the driver launches it for --gpus 2, both ranks on this GPU over gloo, with the
# library-posted exchange carried by the shared-memory stand-in for RCCL.  Prints that leg's record.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/run
mkdir -p $O
export MH_DIST_BACKEND=gloo MAGNET_RCCL_LIBRARY=$PWD/tests/abi/libmock_rccl.so MOCK_RCCL_SLOT_BYTES=134217728
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 2 --steps 2 --warmup 1 --nodes 200000 --edges 4000000 --features 64 --K 3 --strategy ${1:-rows} \
  --cpu-sample-nodes 20000 --cpu-sample-edges 200000 --epoch-nodes 50000 --epoch-edges 1000000 > $O/reh.json 2> $O/reh.err
echo "exit $?"
python3 - <<'PY'
import json
d = [json.loads(l) for l in open('gpurun_out/run/reh.json') if l.startswith('{')][0]
print(d['multi_gpu']['row_variants'].get('extension'))
PY
tail -5 $O/reh.err
