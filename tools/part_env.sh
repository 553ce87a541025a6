# times an eighth of the bench frame and the whole frame for a few settings of the any-hit ranges (development aid)
for kv in "X=1" "RT_AMD_RANGE_GROWTH=4" "RT_AMD_RANGE_GROWTH=6" "RT_AMD_RANGE_GROWTH=8" "RT_AMD_RANGE_GROWTH=4 RT_AMD_RANGE_FIRST=32" "RT_AMD_RANGE_GROWTH=4 RT_AMD_RANGE_FIRST=128" "RT_AMD_RANGE_GROWTH=3 RT_AMD_RANGE_FIRST=32"; do
  echo "== $kv"
  env $kv timeout -k 10 100 python3 tools/part_time.py 8 3 2>&1 | grep "^part" || exit 1
  env $kv timeout -k 10 100 python3 tools/part_time.py 1 0 2>&1 | grep "^part" || exit 1
done
