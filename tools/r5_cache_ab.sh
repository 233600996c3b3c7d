cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_launch_options_do_not_change_results or test_chained_launches" > gpurun_out/cache_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/cache_pytest.log
timeout 900 python tools/gpu_cache_ab.py '{}' '{"waves_per_simd": 4}' 60 61 62 63 > gpurun_out/cache_ab.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/cache_ab.txt
