export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_now; mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bench_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $OUT/f.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bench_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $OUT/w.log 2>&1 || exit 1
python3 scripts/pmc_summary.py $OUT k_primary
