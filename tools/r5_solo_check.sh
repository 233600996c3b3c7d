cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_quality.py tests/test_gpu_parity.py -m gpu -x -q -k "waits_for_every_batch or default_threshold or chained" > gpurun_out/solo_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/solo_pytest.log
python tools/gpu_chain_k.py "{}" "{\"chain_min_rays\": 786432}" 2>&1 | tail -2 | cut -c1-330
python tools/gpu_wait_cost.py 2>&1 | tail -4
