#!/bin/bash
# dev helper (GPU box): merged scatter of a real iteration (tools/time_scatter_real.py), in-tree library against build/variants/librfx_<name>.so ...
# usage: bash tools/r6_knob_ab.sh "office0 scene0000" minseg8 minseg20
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; CFGS=$1; shift
for w in new "$@" new; do
  if [ $w = new ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$w.so; fi
  for c in $CFGS; do
    a=$(ONLY16=1 timeout -k 10 300 python3 $R/tools/time_scatter_real.py $c 2>/dev/null | grep "both" | sed 's/.*: *//')
    echo "$w $c: scatter $a"
  done
done
