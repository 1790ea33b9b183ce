set -u
O=gpurun_out/r05i; rm -rf $O; mkdir -p $O
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1; tail -2 $O/collect.log
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
