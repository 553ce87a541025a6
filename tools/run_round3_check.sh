set -o pipefail
timeout -k 10 120 tools/microbench/valu_rate > gpurun_out/valu_rate_r3.txt 2>&1 && grep "raster4096\|raster2048" gpurun_out/valu_rate_r3.txt
timeout -k 10 600 python -u -m pytest tests/test_bench_obj.py tests/test_gpu_multi.py tests/test_zz_no_alloc.py tests/test_png_textures.py tests/test_jpeg_textures.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5
timeout -k 10 400 python -u bench.py > gpurun_out/bench_r3a.json 2> gpurun_out/bench_r3a.err; echo bench rc=$?; tail -3 gpurun_out/bench_r3a.err
