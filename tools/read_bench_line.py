#!/usr/bin/env python3
"""Human-readable digest of one bench.py JSON line (or of a driver record that holds one under "parsed"): the headline, the parity
statement, the roofline of the dominant kernel and of every streaming stage, the per-rank stage breakdown with max / min over the
ranks, the all-reduce A/B and -- for N = 1 lines -- the strong-scaling projection with each stand-in slab's stages. DESIGN.md 7b
("Reading the first multi-GPU line") says what to look for.
   python tools/read_bench_line.py BENCH_r03.json [more files ...]      or      python bench.py | python tools/read_bench_line.py -"""
import json
import sys


def load(path):
    text = sys.stdin.read() if path == "-" else open(path).read()
    try:
        d = json.loads(text)
    except ValueError:
        d = json.loads([l for l in text.splitlines() if l.startswith("{")][-1])
    return d.get("parsed", d) if isinstance(d, dict) and "metric" not in d else d


def digest(d, name):
    print(f"== {name}: {d.get('n_gpus')} GPU(s), {d.get('steps')} steps after {d.get('warmup')} warm-ups, launched by {d.get('launched_by')}")
    if d.get("value") is None:
        print(f"   UNMEASURED: {d.get('unmeasured')}")
        return
    cfg = d.get("config", {})
    print(f"   {d['value']:.2f} {d['unit']}  ({d['ms_per_step']:.3f} ms per solve, {cfg.get('iterations_per_solve')} iterations, vs A100 x{d.get('vs_baseline') or float('nan'):.2f})"
          f"   transport: {d.get('transport')}; all-reduce: {d.get('allreduce')}; RCCL ranks {d.get('rccl_ranks')}")
    p = d.get("parity_vs_golden", {})
    print(f"   parity vs golden: {'ok' if p.get('ok') else ('n/a: ' + str(p.get('note'))) if not p.get('available') else 'FAILED'}"
          + (f", max rel err {p['max_rel_err']:.2e} over {p['entries_compared']} residuals (bar {p['tolerance']:g})" if p.get("available") else "")
          + f"; ranks agree on history: {d.get('ranks_agree_on_history')}")
    r = d.get("roofline", {})
    print(f"   dominant kernel {r.get('kernel')}: {r.get('avg_launch_ms', 0):.4f} ms = {r.get('achieved', 0) / 1e3:.2f} TB/s = {r.get('frac', 0):.3f} of {r.get('peak', 0) / 1e3:.0f} TB/s"
          + (f"; {r['frac_of_best_stream']:.3f} of the best stream rate of this run ({r['best_stream_gbs_this_run'] / 1e3:.2f} TB/s, {r.get('best_stream_is')})" if r.get("frac_of_best_stream") else "")
          + (f"; fabric traffic {r['traffic'] / 1e9:.2f} GB per launch = {r['traffic'] / r['algorithmic_bytes_per_launch']:.3f} x algorithmic" if r.get("traffic") else "; traffic: n/a"))
    s = r.get("spmv_standalone")
    if s:
        print(f"   SpMV alone (BASELINE metric 1, reference rule): {s['median_ms']:.4f} ms = {s['effective_gbs']:.0f} GB/s effective ({s['effective_gbs_published_formula']:.0f} by the "
              f"published formula = x{s['vs_a100_published']:.2f} the A100), {s['algorithmic_gbs']:.0f} GB/s algorithmic = {s['frac']:.3f} of peak")
    if d.get("degraded"):
        print(f"   DEGRADED: {d['degraded']}")
    for k, v in (r.get("stages") or {}).items():
        if isinstance(v, dict) and "frac" in v:
            print(f"      stage {k:24s} {v['us']:9.1f} us  {v['gbs'] / 1e3:5.2f} TB/s = {v['frac']:.3f}   {v['what']}")
    b = d.get("breakdown")
    if b:
        keys = [k for k in b["max_over_ranks"] if k.endswith("_us")]
        print("   stage breakdown, us per iteration (max / min over ranks):")
        for k in keys:
            print(f"      {k:42s} {b['max_over_ranks'][k]:10.1f} / {b['min_over_ranks'][k]:10.1f}")
        if len(b["per_rank"]) > 1:
            print("   per rank: iteration_us " + " ".join(f"{r_['iteration_us']:.0f}" for r_ in b["per_rank"]) + "; wall ms per step " + " ".join(f"{v:.2f}" for v in d.get("rank_ms_per_step", [])))
    ab = d.get("allreduce_ab")
    if ab:
        print(f"   all-reduce A/B (ms per step): rccl {ab.get('rccl')}, mailbox {ab.get('mailbox')}" + (f"   other leg: {ab['other_leg'].get('error')}" if ab.get("other_leg", {}).get("error") else ""))
    sp = d.get("scaling_probe")
    if sp and "slabs" in sp:
        print(f"   strong-scaling PROJECTION from one GPU ({sp.get('allreduce_path')}):")
        for s in sp["slabs"]:
            if "projected_efficiency_by_allreduce_latency" not in s:
                print(f"      P = {s['gpus']}: a role failed: " + "; ".join(str(r_.get("error")) for r_ in s["roles"]))
                continue
            eff = s["projected_efficiency_by_allreduce_latency"]
            print(f"      P = {s['gpus']}: slowest slab {s['slowest_role_ms_per_solve']:.3f} ms (ideal {s['ideal_ms_per_solve']:.3f}); efficiency at 0 / 10 / 25 / 50 us per all-reduce: "
                  f"{eff['0us']:.3f} / {eff['10us']:.3f} / {eff['25us']:.3f} / {eff['50us']:.3f}")
    c = d.get("cpu_baseline")
    if c:
        print(f"   cpu baseline: {c['value']:.4f} {c['unit']} on {c['cores']} core ({c['kind']}); {c['sample']}")


for path in sys.argv[1:] or ["-"]:
    try:
        digest(load(path), path)
    except Exception as e:  # a record without a parsed line (skipped runs)
        print(f"== {path}: no bench line ({e!r})")
