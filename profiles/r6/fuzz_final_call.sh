# the device differential runs on the round's final kernels (issue priorities on): new seeds
set -u
out=$PWD/gpurun_out/fz2; mkdir -p $out
timeout 600 python tests/fuzz_gpu.py 40000 606 2>&1 | grep -v amdgpu.ids | tail -3 > $out/fuzz_gpu.log
timeout 600 python tests/fuzz_gpu_windows.py 300 606 2>&1 | grep -v amdgpu.ids | tail -2 > $out/fuzz_gpu_windows.log
timeout 600 python tests/fuzz_roll.py 80 606 2>&1 | grep -v amdgpu.ids | tail -2 > $out/fuzz_roll.log
timeout 300 python tests/fuzz_write_streams.py 2>&1 | grep -v amdgpu.ids | tail -2 > $out/fuzz_write.log
tail -n 3 $out/*.log
