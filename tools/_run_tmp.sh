cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r5u_conv.txt
for v in d0 e2 d0 e2; do
  echo "== $v" >> gpurun_out/r5u_conv.txt
  BIDATE_LIB=$GRAFT_REPO_ROOT/fabric_amd/csrc/variants/lib_$v.so BS=1 timeout 300 python tools/bench_conv.py conv >> gpurun_out/r5u_conv.txt 2>&1
done
BIDATE_LIB=$GRAFT_REPO_ROOT/fabric_amd/csrc/variants/lib_e2.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_conv3d.py -x -q -m gpu 2>&1 | tail -3
timeout 900 bash tools/ab_lib.sh d0 e2 > gpurun_out/r5u_ablib.txt 2>&1
cat gpurun_out/r5u_ablib.txt
