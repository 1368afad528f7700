# Round 4, call C: spin tests, repaired multirank / C-consumer tests, mixing tests (GMRES rewrite), the k-point share profile
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_spin.py tests/test_gpu_multirank.py tests/test_gpu_mixing.py tests/test_gpu_scf.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 \
  | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path' > $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
for N in 8 1; do
  timeout 600 python tools/kpoints_share_profile.py $N > $O/kshare_$N.txt 2> $O/kshare_$N.err
  cat $O/kshare_$N.txt
done
timeout 600 python tools/kpoints_share_profile.py 8 --torch-profile > $O/kshare_8_prof.txt 2> /dev/null
tail -45 $O/kshare_8_prof.txt | cut -c1-200
