#!/usr/bin/env python3
"""Randomised cross-check of the 16-bit sparse-conv kernels (runs on the GPU box; a wider net than
tests/test_gpu_conv16.py::test_plan_kernel_equals_table_kernel_on_random_lists, not part of the suite).

Every case draws a neighbour list (empty rows, full rows, shuffled slots, optional row list), widths up to 512, one or two
filter banks, an arithmetic (bf16x3 / f16x2 / f16 activations), forced block shapes (4 / 8 waves, column tiles), and compares
  * the plan-driven kernel with the table-driven one: bit for bit without importance (4e-6 of the range where the slot-range split
    may apply), 2e-6 of the range with it;
  * the plan-driven kernel with a float64 torch evaluation of the same operator (1e-5 of the range; 2e-3 for f16 activations).

usage: python scripts/fuzz_sconv16.py [--cases N] [--seed S]"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "adaptive-surface-reconstruction_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def reference(f, W, Wb, idx, slots, rs, imp, bias, bias_b, relu):
    """float64 on the device: bank a = sum_p W[slot_p]^T f[idx_p]; a bank with importance (the only bank of a single-bank
    call, bank b of a two-bank one) = sum_p imp_p W[slot_p]^T f[idx_p] / sum_p imp_p; then bias, relu"""
    dev = f.device
    v = rs.numel() - 1
    rs, idx, slots = rs.to(dev), idx.to(dev), slots.to(dev)
    rows = torch.repeat_interleave(torch.arange(v, device=dev), rs[1:] - rs[:-1])
    x = f.double()[idx]
    w = imp.double().to(dev)[idx] if imp is not None else None

    def bank(Wk, weighted, b):
        Wk = Wk.double().to(dev)
        out = torch.zeros((v, Wk.shape[2]), dtype=torch.float64, device=dev)
        for k in range(Wk.shape[0]):
            sel = (slots == k).nonzero().flatten()
            if sel.numel():
                xs = x[sel] * w[sel, None] if weighted else x[sel]
                out.index_add_(0, rows[sel], xs @ Wk[k])
        if weighted:
            norm = torch.zeros(v, dtype=torch.float64, device=dev).index_add_(0, rows, w)
            out = torch.where(norm[:, None] != 0, out / torch.where(norm != 0, norm, torch.ones_like(norm))[:, None], out)
        if b is not None:
            out += b.double().to(dev)
        return out

    if Wb is None:
        out = bank(W, imp is not None, bias)
    else:
        out = torch.cat([bank(W, False, bias), bank(Wb, True, bias_b)], 1)
    return (out.clamp_min(0) if relu else out).cpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=2026)
    args = ap.parse_args()
    from asr_hip import ops
    gpu = torch.device("cuda:0")
    ctx = ops.context(gpu)
    g = torch.Generator(device="cpu").manual_seed(args.seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))

    split_rows = ctx.get_option("sconv_split_rows")
    seen = {}
    worst = {"plan_vs_table": 0.0, "plan_vs_f64": 0.0}
    for case in range(args.cases):
        mode = ["bf16x3", "bf16x3", "f16x2", "f16"][ri(0, 3)]
        K = [9, 27, 55][ri(0, 2)]
        cin = [32, 64, 64, 128, 256, 384, 512][ri(0, 6)]
        dual = ri(0, 3) == 0
        cout = 8 * ri(1, 64)
        if dual:
            cout = (cout // 16) * 16 + 8
        cb = 8 if dual else 0
        if cin * (cout + cb) > 512 * 256:  # keep a case in the tens of milliseconds
            cin = 64
        v, num_inp = ri(1, 12000), ri(1, 12000)
        maxlen = min(K, [3, 8, 14, K][ri(0, 3)])
        lens = torch.randint(0, maxlen + 1, (v,), generator=g)
        lens[torch.randint(0, v, (max(1, v // 40),), generator=g)] = K
        total = int(lens.sum())
        slots = torch.cat([torch.randperm(K, generator=g)[:int(n)] for n in lens]) if total else torch.zeros(0, dtype=torch.int64)
        rs = torch.zeros(v + 1, dtype=torch.int64)
        rs[1:] = torch.cumsum(lens, 0)
        idx = torch.randint(0, num_inp, (total,), generator=g)
        ld = cin + [0, 8, 32][ri(0, 2)]
        act = torch.float16 if mode == "f16" else torch.float32
        fbuf = torch.randn((num_inp, ld), generator=g).to(act)
        f = fbuf.to(gpu)[:, :cin]
        W = torch.randn((K, cin, cout), generator=g) * (0.5 / cin ** 0.5)
        Wb = torch.randn((K, cin, cb), generator=g) * (0.5 / cin ** 0.5) if dual else None
        use_imp = dual or ri(0, 2) == 0
        imp = torch.rand(num_inp, generator=g) + 0.05 if use_imp else None
        bias = torch.randn(cout, generator=g) * 0.1 if ri(0, 1) else None
        bias_b = torch.randn(cb, generator=g) * 0.1 if dual and bias is not None else None
        relu = bool(ri(0, 1))
        perm = torch.randperm(v, generator=g).to(torch.int32) if ri(0, 1) else None
        n_rows = max(1, v - ri(0, v // 2)) if perm is not None and ri(0, 2) == 0 else None
        waves = [0, 4, 8][ri(0, 2)]
        ctot_pad = (cout + cb + 15) // 16 * 16
        nts = [n for n in (1, 2, 4, 8) if ctot_pad % (16 * n) == 0]
        force_nt = nts[ri(0, len(nts) - 1)] if ri(0, 2) else 0  # the launcher narrows the column tile on small grids: force the wide ones too
        # the slot-range split of plain 55-slot layers over small grids (plan kernel only) sums a row's slots in three partial
        # sums: another order than the table kernel's -- on in half of the cases
        split_on = ri(0, 1) == 1
        ctx.set_option("sconv_split_rows", split_rows if split_on else 0)
        may_split = split_on and K == 55 and not dual and not use_imp and mode != "f16" and waves == 0 and force_nt == 0
        pk = ops.pack_filters(W.to(gpu), mode, Wb.to(gpu) if dual else None)
        kw = dict(inp_importance=imp.to(gpu) if use_imp else None, normalize=use_imp, relu=relu,
                  bias=bias.to(gpu) if bias is not None else None, bias_b=bias_b.to(gpu) if bias_b is not None else None,
                  row_perm=perm.to(gpu) if perm is not None else None, num_rows=n_rows, cout_b=cb, force_waves=waves, force_nt=force_nt)
        outs = []
        for plan_on in (1, 0):
            ctx.set_option("sconv_plan", plan_on)
            ctx.sconv_variant_counts(reset=True)
            out = torch.full((v, cout + cb), -3.0, device=gpu, dtype=act)
            ops.sparse_conv16(mode, pk, K, cin, cout, f, idx.to(torch.int32).to(gpu), slots.to(torch.uint8).to(gpu), rs.to(gpu),
                              out=out, **kw)
            key = list(ctx.sconv_variant_counts())[0]
            if plan_on:
                seen[key] = seen.get(key, 0) + 1
            outs.append(out.float().cpu())
        ctx.set_option("sconv_plan", 1)
        ctx.set_option("sconv_split_rows", split_rows)
        ref = reference(f.float(), W if mode != "f16" else W.half().float(), Wb if (Wb is None or mode != "f16") else Wb.half().float(),
                        idx, slots, rs, imp, bias, bias_b, relu)
        rows = torch.arange(v) if perm is None else perm.long()[: (n_rows or v)]
        scale = max(1.0, float(ref[rows].abs().max()))
        d_tab = float((outs[0][rows] - outs[1][rows]).abs().max()) / scale
        d_ref = float((outs[0][rows].double() - ref[rows]).abs().max()) / scale
        untouched = True
        if n_rows is not None:
            rest = perm.long()[n_rows:]
            untouched = bool((outs[0][rest] == -3.0).all())
        tol_tab = (4e-6 if may_split else 0.0) if not use_imp else (2e-6 if mode != "f16" else 2e-3)
        tol_ref = 1e-5 if mode != "f16" else 4e-3
        worst["plan_vs_table"] = max(worst["plan_vs_table"], d_tab)
        if mode != "f16":
            worst["plan_vs_f64"] = max(worst["plan_vs_f64"], d_ref)
        ok = d_tab <= tol_tab and d_ref <= tol_ref and untouched
        if not ok or case % 20 == 0:
            print("case %3d %-6s K %2d cin %3d cout %3d+%d v %5d pairs %6d imp %d rows %s waves %d %s: plan-table %.2e plan-f64 %.2e %s"
                  % (case, mode, K, cin, cout, cb, v, total, use_imp, n_rows, waves, key, d_tab, d_ref, "ok" if ok else "FAILED"),
                  flush=True)
        if not ok:
            sys.exit(1)
    print("instances of the plan kernel exercised:")
    for k, c in sorted(seen.items()):
        print("   ", k, c)
    print("all %d cases ok; worst plan-vs-table %.2e, worst plan-vs-f64 (32-bit modes) %.2e" % (args.cases, worst["plan_vs_table"],
                                                                                            worst["plan_vs_f64"]))


if __name__ == "__main__":
    main()
