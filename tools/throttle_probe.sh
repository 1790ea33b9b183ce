mkdir -p gpurun_out/r06j
stat() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
echo "before: $(stat)"
python3 bench.py --no-cpu-baseline --no-scaling-probe --no-compare --no-spmv --steps 20 --warmup 5 > gpurun_out/r06j/quiet.json 2>/dev/null
echo "after quiet run: $(stat)"
python3 tools/read_bench_line.py gpurun_out/r06j/quiet.json | grep -E "iterations/s|direction_update_us  |stage direction"
# 40 busy loops: far beyond the container's 16-CPU quota -> CFS throttles the whole cgroup, the solver's host thread included
for i in $(seq 40); do ( timeout 45 python3 -c "while True: pass" & ) ; done
sleep 2
python3 bench.py --no-cpu-baseline --no-scaling-probe --no-compare --no-spmv --steps 20 --warmup 5 > gpurun_out/r06j/hogged.json 2>/dev/null
echo "after hogged run: $(stat)"
python3 tools/read_bench_line.py gpurun_out/r06j/hogged.json | grep -E "iterations/s|direction_update_us  |stage direction"
sleep 40
