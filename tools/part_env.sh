# times an eighth of the bench frame for a few settings of the shadow phase (development aid)
for kv in "X=1" "RT_AMD_SORT_RAYS=0" "RT_AMD_RANGE_MIN_WORK=2e11" "RT_AMD_RANGE_MIN_WORK=5e11" "RT_AMD_RANGE_FIRST=16" "RT_AMD_RANGE_FIRST=8" "X=2"; do
  echo "== $kv"
  env $kv timeout -k 10 100 python3 tools/part_time.py 8 3 2>&1 | grep "^part" || exit 1
done
