# primary sweep time against the number of triangle chunks (development aid)
for h in 2160 270 1080; do for c in 1 2 4 8 12 16 24 32; do
  echo -n "rows $h chunks $c: "
  RT_AMD_RASTER_FORCE=16,$c timeout -k 10 120 python3 tools/quick_sweep.py --tris 1000000 --w 3840 --h $h --reps 2 2>&1 | grep "rep 1" | cut -c1-110 || exit 1
done; done
