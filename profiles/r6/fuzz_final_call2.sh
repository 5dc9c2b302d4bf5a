# the stream-level differential runs (drop-in against the all-reference build through the shared driver) on the final tree, new seeds
set -u
out=$PWD/gpurun_out/fz4; mkdir -p $out
D=integration/_build/libmzhipdrop.so
timeout 500 python tests/fuzz_lzma_windows.py 250 606 2>&1 | grep -v amdgpu.ids | tail -2 > $out/fuzz_lzma_windows.log
timeout 500 python tests/fuzz_xz_windows.py 200 606 2>&1 | grep -v amdgpu.ids | tail -2 > $out/fuzz_xz_windows.log
timeout 500 python tests/fuzz_wrappers.py 606 3000 $D 2>&1 | grep -v amdgpu.ids | tail -2 > $out/fuzz_wrappers.log
timeout 500 python tests/fuzz_flushed_blocks.py 606 1000 $D 2>&1 | grep -v amdgpu.ids | tail -2 > $out/fuzz_flushed_blocks.log
timeout 400 python tests/fuzz_write_streams.py 80 606 2>&1 | grep -v amdgpu.ids | tail -2 > $out/fuzz_write_streams.log
timeout 400 python tests/fuzz_gpu_windows.py 500 707 2>&1 | grep -v amdgpu.ids | tail -2 > $out/fuzz_gpu_windows.log
tail -n 3 $out/*.log
