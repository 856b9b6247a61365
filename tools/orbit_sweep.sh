for d in 0 4 8 12 16 20 30 45 90; do
  ph=$(python3 -c "print(-90+$d)")
  for cfg in "VV_TILE_LOG2W=5 VV_BRICKED=0" "VV_TILE_LOG2W=3 VV_BRICKED=1 VV_LDS_RESERVE=76000"; do
    echo "delta=$d $cfg"; BENCH_ARGS="--orbit 90,$ph" bash tools/sweep.sh "$cfg" | tail -1
  done
done
# vertical tilt as well
for d in 8 16 30; do
  th=$(python3 -c "print(90-$d)")
  for cfg in "VV_TILE_LOG2W=5 VV_BRICKED=0" "VV_TILE_LOG2W=3 VV_BRICKED=1 VV_LDS_RESERVE=76000"; do
    echo "tilt=$d $cfg"; BENCH_ARGS="--orbit $th,-90" bash tools/sweep.sh "$cfg" | tail -1
  done
done
