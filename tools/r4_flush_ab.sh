#!/bin/bash
# flush_run / flush_chunks through a buffer descriptor (default) against the predicated copies (PC_FLUSH_BUFFER=0), same box
. tools/r4_exp.sh
for v in "" "PYCOLLO_AMD_DEFINES=PC_FLUSH_BUFFER=0" "" "PYCOLLO_AMD_DEFINES=PC_FLUSH_BUFFER=0"; do
  run "$v" "d3 4x12.5k order 8" --problem delta_iii --sections 1785 --order 8 --steps 200 --warmup 30
done
cat $out
