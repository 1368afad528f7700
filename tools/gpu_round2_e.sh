R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
for w in s f; do
  for r in 0 1; do
    REPO=$R PORT=29874 RANK=$r WORLD_SIZE=2 WHICH=$w DFTK_MI_LOBPCG_CHECK=1 MASTER_ADDR=127.0.0.1 timeout 300 python tools/debug_pw4.py > $O/d4_${w}_$r.log 2>&1 &
  done
  wait
  echo "=== $w"; grep -v "amdgpu.ids\|socket.cpp\|Gloo" $O/d4_${w}_0.log | grep -v "rank 1" | tail -70
done
