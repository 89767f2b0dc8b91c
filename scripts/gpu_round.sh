mkdir -p gpurun_out/r03b
(timeout 900 python -m pytest tests/test_ct_kernels_gpu.py -m gpu -q -x --no-header -rN 2>&1 | tail -40) > gpurun_out/r03b/ct.log
(timeout 900 python -m pytest tests -m gpu -q --no-header 2>&1 | tail -40) > gpurun_out/r03b/all.log
tail -25 gpurun_out/r03b/ct.log; echo ======; tail -25 gpurun_out/r03b/all.log
