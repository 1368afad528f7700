R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_lobpcg_blocks.py -q --tb=short -p no:cacheprovider 2>&1 | tail -15 > $O/pytest_blocks.log
tail -8 $O/pytest_blocks.log
for r in 0 1; do
  REPO=$R PORT=29872 RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 timeout 600 python tools/debug_pw2.py > $O/debug_pw2_$r.log 2>&1 &
done
wait
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $O/debug_pw2_0.log | tail -60; tail -5 $O/debug_pw2_1.log
