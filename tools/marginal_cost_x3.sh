#!/bin/bash
# tools/marginal_cost.sh for the split-precision plan (bench.py --precision head_x3): the pipelined step with one stage's
# launches left out of the captured graphs (SIPMASK_DIAG_SKIP).  In this plan a GroupNorm apply reads the conv's f32 output and
# writes the next operand (the in-place ones are skipped together with their conv), so tower / FeatureAlign stages CAN be left
# out: what follows reads the eager run's tensors.  usage: tools/marginal_cost_x3.sh out.txt [repeats]
OUT=${1:-marginal_cost_x3.txt}
REP=${2:-1}
declare -A PAT=(
  [none]=''
  [backbone]='^stem_fused$|conv:backbone\.'
  [fpn]='conv:fpn\.|relu:p6'
  [tower0]='conv:head\.tower0$|gn:(cls|reg)_convs\.0$|split:pyr'
  [tower12]='conv:head\.tower[12]$|gn:(cls|reg)_convs\.[12]$'
  [reg3]='conv:head\.reg_convs\.3$|gn:reg_convs\.3$'
  [gn_all]='^gn:|^gn_stats:'
  [feat_align]='conv:head\.feat_align|gn:feat_align|gn_stats:feat_align|^offset$'
  [mask_branch]='^up:|conv:head\.sip_mask'
  [predictors]='conv:head\.(reg_ctr|cls_cof)'
  [post]='det_select|^nms$|mask_assemble|det_boxes'
)
ORDER=(none backbone fpn tower0 tower12 reg3 gn_all feat_align mask_branch predictors post none)
for r in $(seq 1 $REP); do
  for k in "${ORDER[@]}"; do
    v=$(SIPMASK_DIAG_SKIP="${PAT[$k]}" timeout 300 python bench.py --precision head_x3 --no-extras --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('value_with_launches_skipped') or d['value'], d['ms_per_step'])")
    echo "$k $v" | tee -a "$OUT"
  done
done
