#!/bin/bash
# mixed build A/B on one box (ph-refined Delta III mesh): dispatch order and tile cutting variants
. tools/r4_exp.sh
R="--problem delta_iii --refined 12500 --steps 200 --warmup 30"
run "" "default" $R
run "PYCOLLO_AMD_MIX_CAP_ROWS=24" "cap 24" $R
run "PYCOLLO_AMD_MIX_CAP_ROWS=24 PYCOLLO_AMD_MIX_LPT=1" "cap 24 + longest first" $R
run "PYCOLLO_AMD_MIX_CAP_ROWS=20 PYCOLLO_AMD_MIX_LPT=1" "cap 20 + longest first" $R
run "PYCOLLO_AMD_MIX_CAP_ROWS=28 PYCOLLO_AMD_MIX_LPT=1" "cap 28 + longest first" $R
run "PYCOLLO_AMD_MIX_CAP_ROWS=24 PYCOLLO_AMD_MIX_GROUP=32" "cap 24 + grouped" $R
run "PYCOLLO_AMD_MIX_LPT=1" "longest first" $R
run "" "default again" $R
cat $out
