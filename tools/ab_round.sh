#!/bin/bash
# Same-box pair per config: this round's default step against the previous round's path -- RPO_NO_WS=1 (every prompt-row GEMM
# back on rpo_gemm_nt's 64x64 tiles, split factors 3 / 2) on a -DRPO_TEXT_ATTN_VALU build of the same tree (the text tower's
# attention back on the VALU kernel): the two kernel-path changes of round 5 -- so that profiles/ can show the round's gain
# without box-to-box variance.  Usage (GPU box), after `SRC=attn_text bash tools/build_variant.sh valu -DRPO_TEXT_ATTN_VALU`:
#   bash tools/ab_round.sh > gpurun_out/ab_round.txt
for extra in "" "--model ViT-L/14 --batch 16" "--K 48" "--K 4" "--batch 4" "--batch 8" "--batch 16" "--batch 128" "--dtype f16"; do
  echo "## bench.py $extra"
  python tools/ab_env.py --rounds 2 --steps 60 --extra "$extra" RPO_NO_WS=1,RPO_EARLY_TEXT=0,RPO_HIP_LIB=rpo_amd/build/ab/librpo_valu.so 2>&1 | grep -v amdgpu
done
