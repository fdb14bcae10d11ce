#!/bin/bash
# Rehearsal of the N>1 bench path with N ranks sharing ONE GPU over gloo (iteration counts of the
# slab-wise preconditioner; timings are meaningless here).  usage: weak_scaling_rehearsal.sh cubes N...
cubes=$1; shift
export PHIFEM_DIST_BACKEND=gloo
for n in "$@"; do
  echo "== N=$n cubes=$cubes"
  if [ "$n" = 1 ]; then
    timeout -k 10 500 python bench.py --gpus 1 --cubes $cubes --steps 1 --warmup 1 --no-cpu-baseline | tail -1 > /tmp/ws_$n.json || exit 1
  else
    timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
      --master-port $((29600 + n)) bench.py --gpus $n --cubes $cubes --steps 1 --warmup 1 --no-cpu-baseline | tail -1 > /tmp/ws_$n.json || exit 1
  fi
  python -c "
import json,sys
d=json.load(open('/tmp/ws_$n.json'))
c=d['config']
print('N', d['n_gpus'], 'dofs', c['active_dofs'], 'iterations', c['iterations'], 'relres', c['relres'], 'loop', c.get('dist_loop'), 'ms', round(d['ms_per_step'],1))
"
  sleep 5
done
