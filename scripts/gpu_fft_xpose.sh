cd $GRAFT_REPO_ROOT
for x in 1 2 3; do
  echo "== variant xp$x correctness"
  ECHOPYPE_AMD_LIB=$PWD/echopype_amd/lib/libechopype_amd_xp$x.so python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fft or reference_method_goldens" -x 2>&1 | tail -3
done
PERF_FFT_P64=1 ROUNDS=2 python scripts/perf_fft.py 40000 echopype_amd/lib/libechopype_amd.so echopype_amd/lib/libechopype_amd_xp1.so echopype_amd/lib/libechopype_amd_xp2.so echopype_amd/lib/libechopype_amd_xp3.so 2>&1 | tail -10
