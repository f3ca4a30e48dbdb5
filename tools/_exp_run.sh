source tools/r4_exp.sh
: > $out
for n in 4 6 7 8; do
  K=$(( 12500 / (n - 1) ))
  for T in 896 960 1024; do
  run "PYCOLLO_AMD_TWO_WAVE_TILES=$T" "d3 n$n tiles=$T" --problem delta_iii --sections $K --order $n --steps 200 --warmup 30
  done
done
cat $out
