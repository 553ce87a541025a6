# development aid: the random-shape and stress parity runs with the wave-block shapes of the primary
# sweep and the cell-lattice sweep forced on small images (GPU box, repo root)
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/wave_fuzz.txt}; mkdir -p "$(dirname "$out")"; : > "$out"
for cfg in "X=0:21" "RT_AMD_RASTER_FORCE=4096,3:22" "RT_AMD_RASTER_FORCE=8192,2:23" "RT_AMD_RASTER_FORCE=16384,5:24" \
           "RT_AMD_RASTER_FORCE=32768,1:25" "RT_AMD_RASTER_FORCE=65536,2:26" "RT_AMD_RASTER_FORCE=147456,3:27" \
           "RT_AMD_SHADOW_LATTICE=1:28" "RT_AMD_SHADOW_LATTICE=1 RT_AMD_CELLS_FORCE=2048,3 RT_AMD_RASTER_FORCE=16384,2:29" \
           "RT_AMD_SHADOW_LATTICE=1 RT_AMD_CELLS_FORCE=8,2 RT_AMD_CELLS_FINE=64:30" \
           "RT_AMD_SHADOW_LATTICE=1 RT_AMD_CELLS_PARTS=4:31" "RT_AMD_SHADOW_LATTICE=1 RT_AMD_CELLS_PARTS=8 RT_AMD_CELLS_FINE=4 RT_AMD_CELLS_FORCE=32:32"; do
  env=${cfg%%:*}; seed=${cfg##*:}
  echo "== $env seed $seed" | tee -a "$out"
  env $env timeout -k 10 400 python3 -u tools/fuzz_parts.py --cases 100 --seed $seed 2>&1 | tail -2 | tee -a "$out" || exit 1
done
echo "== stress" | tee -a "$out"
timeout -k 10 500 python3 -u tools/stress_parity.py --seeds 6 2>&1 | tail -4 | tee -a "$out"
echo "== stress, lattice sweep and 4096 x 256 blocks forced" | tee -a "$out"
RT_AMD_SHADOW_LATTICE=1 RT_AMD_RASTER_FORCE=16384,2 timeout -k 10 500 python3 -u tools/stress_parity.py --seeds 4 2>&1 | tail -4 | tee -a "$out"
