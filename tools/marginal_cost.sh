#!/bin/bash
# Marginal cost of a stage INSIDE the pipelined step: bench.py's timed loop with the stage's launches left out of the
# captured graphs (SIPMASK_DIAG_SKIP, engine._run_steps).  Interleaved repeats; prints img/s and ms per step.
# Only stages whose removal leaves the DATA of the following stages plausible are listed: the buffers keep the eager run's
# values, so what follows a skipped backbone / FPN / predictor launch reads stale but valid tensors.  Skipping the towers, the
# GroupNorm applies or FeatureAlign does not qualify -- the in-place GroupNorm passes then renormalise stale tensors every
# replay, the scores drift to NaN and the post-processing tail changes (measured nonsense: 'slower without FeatureAlign').
# usage: tools/marginal_cost.sh out.txt [repeats]
OUT=${1:-marginal_cost.txt}
REP=${2:-2}
declare -A PAT=(
  [none]=''
  [stem]='^nhwc$|conv:stem|^maxpool$|^stem_fused$'
  [layer1]='conv:backbone\.layer1\.'
  [layer2]='conv:backbone\.layer2\.'
  [layer3]='conv:backbone\.layer3\.'
  [layer3_1x1]='conv:backbone\.layer3\..*(conv1|conv3|downsample)$'
  [layer4]='conv:backbone\.layer4\.'
  [fpn]='conv:fpn\.|relu:p6'
  [fpn_small]='conv:fpn\.(lat2|lat1|p6|p7)|relu:p6'
  
  
  
  [mask_branch]='^up:|conv:head\.sip_mask'
  [predictors]='conv:head\.(reg_ctr|cls_cof)'
  [post]='det_select|^nms$|mask_assemble|det_boxes'
)
ORDER=(none stem layer1 layer2 layer3 layer3_1x1 layer4 fpn fpn_small mask_branch predictors post none)
for r in $(seq 1 $REP); do
  for k in "${ORDER[@]}"; do
    v=$(SIPMASK_DIAG_SKIP="${PAT[$k]}" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('value_with_launches_skipped') or d['value'], d['ms_per_step'])")
    echo "$k $v" | tee -a "$OUT"
  done
done
