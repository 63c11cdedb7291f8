#!/bin/bash
# A/B of the contact leg with and without set-up reuse: three runs each, alternating.
for i in 1 2 3; do
  python bench.py --contact-only --steps 20 --warmup 5 > gpurun_out/ct_ab_reuse_$i.json 2> gpurun_out/ct_ab_reuse_$i.err || exit 1
  MPM_CT_NO_REUSE=1 python bench.py --contact-only --steps 20 --warmup 5 > gpurun_out/ct_ab_noreuse_$i.json 2> gpurun_out/ct_ab_noreuse_$i.err || exit 1
done
