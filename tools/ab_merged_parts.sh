#!/bin/bash
# usage (GPU box): tools/ab_merged_parts.sh <world>  -- what the parts of a merged slice cost (timing experiments: the dbg runs give wrong results):
# AE_SL_DBG 16 = no wipes of the dependency words, 32 = no loads of them (no event waits), AE_SL_NO_PREMARK = the next slice's words filled by a launch of its own
W=${1:-8}
run() { echo "$1:"; shift; env AE_DEBUG_KNOBS=1 AE_SL_SHARD_TILE=1 "$@" timeout 600 python tools/run_rank_share.py 11000000 $W 4 2>&1 | grep "RESULT" | tail -1; }
run "default"
run "no wipes" AE_SL_DBG=16
run "no dependency loads" AE_SL_DBG=32
run "neither" AE_SL_DBG=48
run "marks as a launch of their own" AE_SL_NO_PREMARK=1
run "overflow class in every slice" AE_SL_OV_EVERY_SLICE=1
run "one launch per class" AE_SL_NO_MERGE=1
