set -u
mkdir -p gpurun_out/r05a
python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r05a/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05a/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r05a/smoke.log
for R in 1:0 2:1 4:1 8:0 8:3; do
  python3 tools/slab_attribution.py --child 20000 5 $R > gpurun_out/r05a/new_$R.json 2>gpurun_out/r05a/new_$R.err
  SLAB_OPTIONS=reduce_one_launch=0 python3 tools/slab_attribution.py --child 20000 5 $R > gpurun_out/r05a/old_$R.json 2>gpurun_out/r05a/old_$R.err
done
python3 - <<'PY'
import json,glob
for R in ("1:0","2:1","4:1","8:0","8:3"):
    for k in ("new","old"):
        try:
            rec=json.load(open(f"gpurun_out/r05a/{k}_{R}.json"))[0]
        except Exception as e:
            print(R,k,"failed",e); continue
        st=rec["stage_us"]
        import numpy as np
        print(f"{R} {k}: solve {rec['event_ms']:.3f} ms wall {rec['wall_ms']:.3f}; spmv in-loop {np.mean(rec['spmv_inloop_us']):.1f}; stages: interior {st['spmv_interior_us']}, boundary {st['halo_wait_and_boundary_rows_us']}, redA {st['reduce_pAp_and_allreduce_us']}, upd_r {st['update_r_us']}, redB {st['reduce_rr_allreduce_and_scalar_step_us']}, dir {st['direction_update_us']}, gap {st['gap_before_next_iteration_us']}, iter {st['iteration_us']}")
PY
