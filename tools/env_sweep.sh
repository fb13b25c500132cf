#!/bin/bash
# HIP runtime knobs vs the per-kernel cost of a replayed dependent chain (tools/chain_overlap.py): one setting per line.
#   bash tools/env_sweep.sh > gpurun_out/env_sweep.txt
cd "$(dirname "$0")/.."
run() {
  echo "== $*"
  env "$@" timeout -k 10 120 python tools/chain_overlap.py 4096 2>&1 | grep "rows" | cut -c1-110
}
run X=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run ROC_USE_FGS_KERNARG=0
run ROC_USE_FGS_KERNARG=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run GPU_MAX_HW_QUEUES=8
run AMD_DIRECT_DISPATCH=0
run ROC_SKIP_KERNEL_ARG_COPY=1
run DEBUG_HIP_DYNAMIC_QUEUES=0
run DEBUG_HIP_DYNAMIC_QUEUES=1
