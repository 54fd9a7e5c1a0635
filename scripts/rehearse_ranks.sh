#!/bin/bash
# Rehearsal of bench.py's N > 1 code path with several ranks on ONE GPU (gloo backend; RCCL refuses to share a
# device between ranks): block-cyclic staged / inplace / p2p gathers, slab fallback, padded gather.
# usage: scripts/rehearse_ranks.sh   (<= 4 ranks at a time)
port=29550
run() {   # nranks dims extra...
  n=$1; d=$2; shift 2
  port=$((port+1))
  out=$(timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus $n --backend gloo --dims $d --steps 2 --warmup 1 --cpu-rows 0 "$@" 2>/dev/null | tail -1)
  python - "$n" "$d" "$*" "$out" <<'PY'
import sys, json
n, d, extra, line = sys.argv[1:5]
try:
    j = json.loads(line)
    print(f"ranks {n} dims {d} {extra:28s} -> {j['config']['parallelism'][:60]:60s} selfcheck {j['selfcheck']} exchange {j['exchange']['mode']}")
except Exception as e:
    print(f"ranks {n} dims {d} {extra}: FAILED ({e}) {line[:200]}")
    sys.exit(1)
PY
}
run 2 63 --gather staged && run 4 63 --gather inplace && run 4 63 --gather p2p && run 2 61 && run 3 63 && run 3 47 --gather p2p --mode vdw
