# times bench.py's default frame with each library variant (development aid)
for n in default "$@"; do
  if [ $n = default ]; then lib=raytracer_amd/librt_amd.so; else lib=variants/librt_$n.so; fi
  echo "== $n"
  RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 2 2>/dev/null | python3 tools/bench_line.py || exit 1
done
