# development aid: the random-shape and stress parity runs with wide lanes of the primary sweep forced on small images
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/lattice_fuzz.txt}; : > "$out"
for cfg in "X=0:11" "RT_AMD_RASTER_FORCE=256,3:12" "RT_AMD_RASTER_FORCE=128,2:13" "RT_AMD_RASTER_FORCE=64,5:14" "RT_AMD_RASTER_FORCE=256,1,2:15" "RT_AMD_RASTER_FORCE=1024,2:17" "RT_AMD_RASTER_FORCE=512,3:18" "RT_AMD_RASTER_FORCE=2048,2:19" "RT_AMD_CULL=1:16"; do
  env=${cfg%%:*}; seed=${cfg##*:}
  echo "== $env seed $seed" | tee -a "$out"
  env $env timeout -k 10 400 python3 -u tools/fuzz_parts.py --cases 100 --seed $seed 2>&1 | tail -2 | tee -a "$out" || exit 1
done
echo "== stress" | tee -a "$out"
timeout -k 10 500 python3 -u tools/stress_parity.py --seeds 6 2>&1 | tail -4 | tee -a "$out"
