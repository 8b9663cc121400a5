cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz_packed.py -q -x 2>&1 | tail -15
