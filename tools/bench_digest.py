#!/usr/bin/env python3
"""Digest of bench.py JSON lines for the text files under profiles/ (value, kernel time, parity, ladder, exchange).

usage: bench_digest.py "<command shown>"=<file with the JSON line> ..."""
import ast
import json
import sys


def obj(v):
    if isinstance(v, str) and v[:1] in "{[":
        try:
            return ast.literal_eval(v)
        except Exception:
            return v
    return v


def digest(cmd, path):
    line = [l for l in open(path) if l.startswith("{")][-1]
    d = json.loads(line)
    c, r = d["config"], d["roofline"]
    print(f"$ {cmd}")
    print(f"  value {d['value'] / 1e9:.4f} G frames/s ({c.get('frames_per_gpu')} frames per GPU, scaling {d.get('scaling')}), ms_per_step {d['ms_per_step']:.4f}, "
          f"n_gpus {d['n_gpus']}, kernel_ms {r['kernel_ms']:.4f}"
          f" (+ accumulate kernel {r.get('accumulate_kernel_ms', 0.0):.4f}), roofline.frac {r['frac']:.4f}"
          + (f" ({r['ksteps_per_pair']:.3f} k-steps per pair counted over the timed passes, flagged {r['flagged_fraction']:.4f};"
             f" one-stage equivalent {r['one_stage_equivalent']['frac_of_peak']:.3f})" if r.get("two_stage") else ""))
    print(f"  sweep launches in the timed region {json.dumps(c.get('timed_sweep_launches'))}")
    par = obj(c.get("parity"))
    if par:
        print("  parity " + json.dumps(par)[:300])
    col = obj(c.get("collective"))
    print("  collective " + (json.dumps(col)[:600] if col else "null"))
    e2e = obj(c.get("learn_end_to_end"))
    if isinstance(e2e, dict):
        lv = e2e["levels"]
        print(f"\n  ladder {e2e['seconds'] * 1e3:.2f} ms, passes {e2e['passes_per_level']}; per level kernel ms / step ms per pass: "
              + ", ".join(f"M={l['M']}: {l['kernel_ms']:.3f} / {l['step_ms']:.3f}" for l in lv))
        if any("allreduce_us_per_call" in l for l in lv):
            print("\n  per level allreduce_us_per_call: " + ", ".join(f"M={l['M']}: {l['allreduce_us_per_call']:.1f}" for l in lv
                                                                      if "allreduce_us_per_call" in l))
    an = obj(c.get("weak_scaling_anchor"))
    if isinstance(an, dict) and "value" in an:
        print(f"\n  anchor ({an['frames']} frames on one GPU): value {an['value'] / 1e9:.4f} G frames/s, ms_per_step {an['ms_per_step']:.4f}, "
              f"kernel_ms {an['kernel_ms']:.4f}, {an['ksteps_per_pair']:.3f} k-steps per pair")
        al = an["learn_end_to_end"]
        print(f"  anchor ladder {al['seconds'] * 1e3:.2f} ms, passes {al['passes_per_level']}; per level kernel ms / step ms per pass: "
              + ", ".join(f"M={l['M']}: {l['kernel_ms']:.3f} / {l['step_ms']:.3f}" for l in al["levels"]))
    elif an:
        print("\n  anchor " + json.dumps(an)[:300])
    rb = obj(c.get("robustness"))
    if isinstance(rb, dict):
        print(f"\n  robustness (M = {c.get('codebook_size')} level on {rb['frames']} frames; all equal the plain sweep: {rb['all_equal_plain_sweep']}; "
              f"worst product / fastest variant: {rb.get('worst_product_over_fastest')}):")
        for g in rb["generators"]:
            if "error" in g:
                print(f"    {g['generator']:22s} ERROR {g['error']}")
                continue
            print(f"    {g['generator']:22s} r0 {g['mean_r0']:8.2f}  flagged(first pass) {g['flagged_fraction_first_pass']}  equals plain {g['equals_plain_sweep']}  "
                  f"step ms per pass {g['step_ms_per_pass']}  fastest {g['fastest']}  product/fastest {g['product_over_fastest']:.3f}  decided {g['product'].get('decided')}")
            for k in ("product", "one_stage_after_first_pass", "sorted_one_stage", "sorted_two_stage", "round4_kernel", "plain_sweep"):
                if k not in g:
                    continue
                v = g[k]
                print(f"        {k:17s} kernel ms {v['kernel_ms']}  step ms {v['step_ms']}  {v['sweeps']}  uncertified {v['uncertified_fraction']}")
    qz = obj(c.get("quantize"))
    if isinstance(qz, dict):
        rq = qz["roofline_quantize"]
        print(f"\n  quantize ({qz['frames']} frames, M = {qz['codebook_size']}): {qz['frames_per_sec_device_resident'] / 1e9:.3f} G frames/s, kernels {qz['kernel_ms']:.3f} ms, "
              f"roofline_quantize.frac {rq['frac']:.3f} ({rq['achieved']:.0f} TF executed)")
    s16 = obj(c.get("strong_scaling_16M"))
    if isinstance(s16, dict):
        print(f"\n  16 M frames on one GPU: M = 1024 level {s16['level_ms_per_pass']:.3f} ms per pass (kernels "
              f"{s16['level_kernel_ms_per_pass']:.3f}), ladder {s16.get('ladder_seconds', float('nan')):.4f} s")
    q = c.get("quantize_frames_per_sec_device_resident")
    if q:
        print(f"\n  device-resident quantize at M = 1024: {q / 1e9:.3f} G frames/s")
    sc = obj(c.get("small_corpus"))
    if isinstance(sc, dict):
        print("\n  small corpus " + json.dumps({k: v for k, v in sc.items() if k != "levels"}))
        print("  small corpus per level kernel us / step us per pass: "
              + ", ".join(f"M={l['M']}: {l['kernel_us_per_pass']} / {l['step_us_per_pass']}" for l in sc["levels"]))
    cb = d.get("cpu_baseline")
    if cb:
        print(f"\n  cpu_baseline {cb['value'] / 1e6:.2f} M frames/s on {cb['cores']} threads ({cb['kind']}): {cb['sample']}")
    print()


for a in sys.argv[1:]:
    cmd, path = a.rsplit("=", 1)
    digest(cmd, path)
