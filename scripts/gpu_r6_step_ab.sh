#!/bin/bash
# the c4 training step (graphed, batch 64 x 512 frames) with one environment switch at 0 and 1, several rounds in one session
# usage: gpu_r6_step_ab.sh VAR [rounds]
var=$1; rounds=${2:-3}
for r in $(seq 1 $rounds); do
  for v in 1 0; do
    env $var=$v python scripts/gpu_graph_train.py 2>&1 | grep -i "graph\|eager" | tail -2 | sed "s/^/$var=$v  /"
  done
done
