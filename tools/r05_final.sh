set -u
O=gpurun_out/r05h; rm -rf $O; mkdir -p $O
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1; tail -2 $O/collect.log
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
for G in 10000 15000; do python3 bench.py --grid $G --no-cpu-baseline > $O/bench_$G.json 2>$O/bench_$G.err; echo "bench $G rc=$?"; done
