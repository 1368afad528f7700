# HBM traffic of every kernel family of the benchmark workload (TCC byte counters, separate passes per
# the gfx950 slot limits).  Writes gpurun_out/r01_pmc_traffic.json (copied to profiles/ by hand).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf --output-format csv -- python $R/bench.py --no-cpu-baseline --prof-all > /tmp/bench_f.json 2>/dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw --output-format csv -- python $R/bench.py --no-cpu-baseline --prof-all > /tmp/bench_w.json 2>/dev/null
python $R/tools/pmc_to_traffic.py /tmp/pf/pf_counter_collection.csv /tmp/pw/pw_counter_collection.csv /tmp/bench_f.json $R/gpurun_out/r01_pmc_traffic.json si4x4x4_ecut30
python $R/tools/pmc_summary.py /tmp/pf/pf_counter_collection.csv FETCH_SIZE 24 > $R/gpurun_out/pmc_bench_fetch.txt
python $R/tools/pmc_summary.py /tmp/pw/pw_counter_collection.csv WRITE_SIZE 24 > $R/gpurun_out/pmc_bench_write.txt
cat $R/gpurun_out/r01_pmc_traffic.json
