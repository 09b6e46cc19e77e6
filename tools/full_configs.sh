#!/bin/bash
# the BASELINE configs at their full sizes on one MI355X (rows + engine copy fit in 288 GB), one JSON line each -> gpurun_out/full_configs.jsonl
set -e
out=gpurun_out/full_configs.jsonl; : > $out
run() { timeout -k 10 900 python tools/bench_configs.py "$@" 2>gpurun_out/full_cfg.err | tail -1 >> $out; tail -1 $out | cut -c1-260; }
run c4 5000000
HX_ITER_QUERIES=6000 HX_C5_QUERIES=10000 run c5 20000000
run c3 10000000
