# The driver's N = 2 bench command on a ONE-GPU box: both ranks on cuda:0, collectives host-staged over gloo (RCCL wants one
# device per rank).  Not a timing -- the point is the parity block of the N = 2 line at the BASELINE size (sharded fixed point
# == one-rank fixed point, tests/golden/device_one_rank_energies.json).  Usage: bench_two_ranks_one_gpu.sh <out-prefix> [bench args]
OUT=$1; shift
PORT=$((29500 + RANDOM % 400))
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT DFTK_MI_BENCH_BACKEND=gloo DFTK_MI_BENCH_DEVICE=0 \
    timeout 1500 python bench.py --gpus 2 --no-cpu-baseline "$@" > ${OUT}_rank$r.json 2> ${OUT}_rank$r.err &
done
wait
tail -c 400 ${OUT}_rank0.err
