cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 bash tools/ab_lib.sh h0 hn > gpurun_out/r5r_ablib.txt 2>&1
cat gpurun_out/r5r_ablib.txt
