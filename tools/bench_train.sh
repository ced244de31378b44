#!/bin/bash
# bash tools/bench_train.sh TAG [ENV=VAL ...]: the training-step bench line under the given environment
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 python bench.py --no-cpu-baseline --workload train_dagnet_bsk_8192 --steps 5 --warmup 2 > gpurun_out/train_$TAG.json 2> gpurun_out/train_$TAG.err || { tail -5 gpurun_out/train_$TAG.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/train_$TAG.json')); print('$TAG', '%.1f ms/step' % d['ms_per_step'], 'loss %.6g' % d['loss'])"
