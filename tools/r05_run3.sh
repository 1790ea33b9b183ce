set -u
export TMPDIR=/tmp
O=gpurun_out/r05c; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
for ROLE in "20000 8 3" "20000 8 0" "20000 4 1" "20000 2 1" "10000 8 3" "10000 4 1"; do
  F="$O/ab_wait_$(echo $ROLE | tr ' ' '_').txt"
  python3 tools/ab_loop_options.py halo_wait_first $ROLE 8 > "$F" 2>&1
  grep -h "halo_wait_first=\|1 vs 0\|^grid" "$F"
done
