#!/bin/bash
# usage (GPU box): tools/ab_class_caps.sh <world> [caps...]  -- a rank's share of configs[3]'s graph (tools/run_shard_time.py, tile of negatives) with palettes of several widths
W=${1:-8}; shift
for cap in ${@:-11 15 19 24}; do
  echo "palette $cap:"
  AE_DEBUG_KNOBS=1 AE_SL_SHARD_TILE=1 AE_SL_CLASS_CAP=$cap AE_CE_PROF=1 timeout 600 python tools/run_shard_time.py $W 2>&1 | grep "world\|colouring" | tail -2
done
