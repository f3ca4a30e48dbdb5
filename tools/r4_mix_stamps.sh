#!/bin/bash
# per-order wave life of the mixed build on the ph-refined Delta III mesh (PC_STAMPS object built beforehand)
mkdir -p gpurun_out
PYCOLLO_AMD_DEFINES=PC_STAMPS timeout -k 10 400 python tools/stamps.py --problem delta_iii --refined 12500 --reps 5 > gpurun_out/mix_stamps.txt 2>&1
tail -40 gpurun_out/mix_stamps.txt
