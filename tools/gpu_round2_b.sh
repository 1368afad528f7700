R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
python -c "import dftk_jl_amd as d; print(d.load_library().dftk_mi_version())" > $O/version.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_lobpcg_blocks.py -q --tb=short -s -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/pytest_blocks.log
tail -40 $O/pytest_blocks.log
for r in 0 1; do
  REPO=$R PORT=29871 RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 timeout 300 python tools/debug_pw.py > $O/debug_pw_$r.log 2>&1 &
done
wait
tail -30 $O/debug_pw_0.log; tail -5 $O/debug_pw_1.log
