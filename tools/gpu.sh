#!/bin/bash
# development helper: build here (stop on failure), then run a command on the GPU box.  usage: tools/gpu.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python -m annembed_amd.build > /tmp/ae_build.log 2>&1 || { grep -B2 -A8 "error" /tmp/ae_build.log | head -60; echo BUILD FAILED; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
