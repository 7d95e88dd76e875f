#!/bin/bash
# A/B of mlp_forward_kernel's forms on ONE box (NRF_MLP_FORM: 30 = three workgroups per CU, prefetch buffers rotated by register
# copies -- rounds 1-4; 20 = two per CU, copies; 21 = two per CU, rotated by name: loads really two chunks ahead), alternating.
# usage: scripts/mlp_form_ab.sh  -> gpurun_out/mlp_form_ab.txt
set -eo pipefail
mkdir -p gpurun_out
: > gpurun_out/mlp_form_ab.txt
for rep in 1 2 3; do
  for form in ${FORMS:-30 20 21}; do
    echo -n "NRF_MLP_FORM=$form  " >> gpurun_out/mlp_form_ab.txt
    NRF_MLP_FORM=$form python3 scripts/mlp_steady.py 2>/dev/null | tail -1 >> gpurun_out/mlp_form_ab.txt
  done
done
cat gpurun_out/mlp_form_ab.txt
