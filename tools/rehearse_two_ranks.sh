#!/bin/bash
# Two ranks of `bench.py --gpus 2` sharing ONE GPU over gloo at a size where everything of the N > 1 path is live:
# --nsza 30 = 2730 rows = 16.6 GB of LUT, windows of 8.3 GB (placed by gort_lut_alloc's scan), in-place all-gather,
# parity of rows the other rank computed, config-5 block with 40 members.  A rehearsal of the code path, not a measurement.
cd "$(dirname "$0")/.."
timeout -k 10 ${TIMEOUT:-900} python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29549 \
    bench.py --gpus 2 --steps 5 --warmup 2 --nsza 30 --rehearse --c5-members 40 --c5-chunk 10 --no-cpu-baseline --sustain-s 0.5
