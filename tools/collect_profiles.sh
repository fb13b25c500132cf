#!/bin/bash
# Everything profiles/rNN_* is made of, in one GPU-box call:   bash tools/collect_profiles.sh r04   (writes gpurun_out/<tag>_profiles/)
#   default bench line, rocprofv3 kernel stats of the north-star bench and of the Point-M2AE step, the two --pmc passes of each
#   (separate runs: no tracing beside counter collection), their per-launch summaries and the hash of csrc/ they were measured on.
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${tag}_profiles
mkdir -p $out
cd $R
python bench.py > $out/${tag}_bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
GM3D_FORCE_DIST=1 python bench.py --no-secondary --no-cpu-baseline > $out/${tag}_bench_forced_dist.json 2> $out/bench_dist.err; echo "dist rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ns -- python3 $R/bench.py --steps 45 --warmup 3 --no-cpu-baseline --no-secondary > $out/${tag}_graph_bf16_bench.json 2> $out/prof_ns.err; echo "prof ns rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_m2ae -- python3 $R/tools/bench_m2ae.py --steps 10 --warmup 3 > $out/${tag}_bench_m2ae.json 2> $out/prof_m2ae.err; echo "prof m2ae rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/pmc_$c --output-format csv -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $out/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  rocprofv3 --pmc $c -d $out/m2ae_pmc_$c --output-format csv -- python3 $R/tools/bench_m2ae.py --steps 2 --warmup 1 > $out/m2ae_pmc_$c.log 2>&1; echo "m2ae pmc $c rc=$?"
done
cd $R
python tools/pmc_summary.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/${tag}_pmc_fetch_write_per_launch.json
python tools/pmc_summary.py $out/m2ae_pmc_FETCH_SIZE $out/m2ae_pmc_WRITE_SIZE > $out/${tag}_m2ae_pmc_fetch_write_per_launch.json
python -c "import bench, json; print(json.dumps({'sha16': bench.kernel_sources_sha16()}))" > $out/${tag}_pmc_sources.json
cp $(find $out/prof_ns -name "*kernel_stats.csv" | head -1) $out/${tag}_graph_bf16_kernel_stats.csv
cp $(find $out/prof_m2ae -name "*kernel_stats.csv" | head -1) $out/${tag}_m2ae_kernel_stats.csv
find $out -name "*counter_collection.csv" -delete
find $out -name "*kernel_trace.csv" -delete
ls -la $out | head -30
