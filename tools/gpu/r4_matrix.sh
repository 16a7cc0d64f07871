cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/matrix
# the GPU parity suite under every kernel-selection switch: the bytes must not depend on which kernel runs
for e in IQGPU_STEAL=1 IQGPU_NO_S2=1 IQGPU_NO_FAT=1 IQGPU_FAT=1 IQGPU_NO_FUSED_MOVE=1 IQGPU_FORCE_FAT=1; do
  n=${e%%=*}
  env $e timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider > gpurun_out/matrix/$n.log 2>&1; rc=$?
  echo "$e rc=$rc $(tail -1 gpurun_out/matrix/$n.log)"
  grep -E "^FAILED" gpurun_out/matrix/$n.log | cut -c1-160 | head -8
  [ $rc -le 1 ] || exit 1
done
