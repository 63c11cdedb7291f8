#!/bin/bash
# A/B of the first batch of a contact solve (MPM_CT_BATCH=first,next): default (as many iterations as the last solve took)
# against short first batches; four runs each, interleaved.
for i in 1 2 3 4; do
  for v in default 2,1 4,1; do
    if [ $v = default ]; then unset MPM_CT_BATCH; else export MPM_CT_BATCH=$v; fi
    python bench.py --contact-only --steps 20 --warmup 5 > gpurun_out/ct_batch_${v/,/_}_$i.json 2>/dev/null || exit 1
  done
done
