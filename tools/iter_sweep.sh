#!/bin/bash
# tuning sweep of the device-resident iterative scan (C5 shape): LDS head of the discarded heap x resident workgroups per CU
for cfg in "512 14" "512 12" "512 10"; do
  set -- $cfg
  echo "== disc_lds $1 per_cu $2"
  HX_DISC_LDS=$1 HX_ITER_PER_CU=$2 HX_ITER_QUERIES=6000 HX_C5_QUERIES=6000 timeout -k 10 300 python tools/bench_configs.py c5 1000000 clustered 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['iterative_relaxed']['qps'], d['iterative_relaxed']['recall_at_10'], d['fused']['redone'])" || exit 1
done
