cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04_final_pytest.txt 2>&1
tail -3 gpurun_out/r04_final_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_final_smoke.txt 2>&1; tail -1 gpurun_out/r04_final_smoke.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err
S="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_serial -- python bench.py $S > gpurun_out/r04_prof_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_concurrent -- python bench.py $S > gpurun_out/r04_prof_concurrent.log 2>&1
STEP=4 python tools/timeline_overlap.py $(ls gpurun_out/r04_prof_concurrent/*/*kernel_trace.csv | head -1) > gpurun_out/r04_timeline_c4.txt 2>&1
cp $(ls gpurun_out/r04_prof_serial/*/*kernel_stats.csv | head -1) gpurun_out/r04_final_kernel_stats_serial.csv
cp $(ls gpurun_out/r04_prof_concurrent/*/*kernel_stats.csv | head -1) gpurun_out/r04_final_kernel_stats_concurrent.csv
rm -rf gpurun_out/r04_prof_serial gpurun_out/r04_prof_concurrent
timeout 300 python tools/solve_c4_lp.py > gpurun_out/r04_c4_lp_end_to_end.txt 2>&1
HEADLINE=1 timeout 400 python tools/solve_c4_lp.py > gpurun_out/r04_headline_lp_end_to_end.txt 2>&1
head -c 300 gpurun_out/r04_final_bench.json
