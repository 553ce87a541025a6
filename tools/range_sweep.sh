# frame time against the any-hit range schedule (development aid)
for cfg in "2.0 64" "1.5 64" "1.3 64" "2.0 256" "1.5 256" "1.3 128" "1.2 256"; do
  set -- $cfg
  echo -n "growth $1 first 1/$2: "
  RT_AMD_RANGE_GROWTH=$1 RT_AMD_RANGE_FIRST=$2 timeout -k 10 200 python3 tools/band_scaling.py --ranks 8 --detail 2>&1 | grep -E "part 0/1|part 3/8" | sed 's/.*shadow/shadow/' | tr '\n' ' ' || exit 1
  echo
done
