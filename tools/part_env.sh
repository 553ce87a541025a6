# times an eighth of the bench frame for a few settings of the shadow phase (development aid)
for kv in "X=1" "RT_AMD_RAY_ORDER=u" "X=2" "RT_AMD_RAY_ORDER=u"; do
  echo "== $kv"
  env $kv timeout -k 10 100 python3 tools/part_time.py 8 3 2>&1 | grep "^part" || exit 1
  env $kv timeout -k 10 100 python3 tools/part_time.py 4 1 2>&1 | grep "^part" || exit 1
done
