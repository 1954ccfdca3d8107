mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rf -s 2>&1 | grep -v "^\[wetts" > gpurun_out/pytest_gpu_full.log
tail -15 gpurun_out/pytest_gpu_full.log
grep -n "vs f32: rel rms\|^OK \|^FAIL " gpurun_out/pytest_gpu_full.log | head -20
