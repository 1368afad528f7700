R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
echo "=== 1 rank, unsharded only, poison"; REPO=$R WORLD_SIZE=1 RANK=0 ORDER=f DFTK_MI_POISON=1 timeout 300 python tools/debug_pw3.py 2>&1 | grep -v "amdgpu.ids" | tail -12
for mode in "sf 0" "sf 1" "fs 0"; do
  set -- $mode
  echo "=== 2 ranks order=$1 poison=$2"
  for r in 0 1; do
    if [ "$2" = "1" ]; then export DFTK_MI_POISON=1; else unset DFTK_MI_POISON; fi
    REPO=$R PORT=29873 RANK=$r WORLD_SIZE=2 ORDER=$1 MASTER_ADDR=127.0.0.1 timeout 300 python tools/debug_pw3.py > $O/d3_$1_$2_$r.log 2>&1 &
  done
  wait
  grep -v "amdgpu.ids\|socket.cpp\|Gloo" $O/d3_$1_$2_0.log | tail -16
done
