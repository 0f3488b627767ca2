#!/usr/bin/env python3
"""Per-shape roofline table from tools/gemm_lab output: for every encoder GEMM shape the algorithmic HBM bytes, the two
floors (16-bit MFMA dense peak 2.5 PFLOP/s; HBM 8 TB/s peak and the 6.3 TB/s the guide measures as achievable), which
one binds, and the measured fraction of the binding floor. TFLOP/s are always ALGORITHMIC (2 M N K).

  gemm_shapes_report.py lab.txt                        one table; the lab file's header says which operand mode it timed:
        bf16     one 16-bit plane per operand, one MFMA term
        fp16x3   two planes per 16-bit operand / output, three terms executed per algorithmic product
        fp16x2   two terms (SplitArgs::terms = 2): ONE activation plane read, two weight planes, the GELU output one plane
  gemm_shapes_report.py lab_x3.txt lab_x2.txt TAGS     the FP16X3M table: the shapes named by TAGS ("qkv.s2,fc1.s2,fc2.s2", the
        syntax of Engine.set_op_terms) are taken from the two-term file, the rest from the three-term file; the whole-encoder
        line is then the mixed mode's
"""
import sys

PEAK_TF, HBM_PEAK, HBM_ACH = 2500.0, 8.0e12, 6.29e12
LAYERS = {0: 2, 1: 2, 2: 18, 3: 2}


def parse(path):
    head = open(path).read(4000)
    mode = "fp16x2" if "fp16x2" in head else "fp16x3" if "fp16x3" in head else "bf16"
    rows = {}
    for ln in open(path):
        p = ln.split("|")
        if len(p) < 4 or p[0].startswith("shape"):
            continue
        h = p[0].split()
        name, epi, M, N, K = " ".join(h[:-4]), int(h[-4]), int(h[-3]), int(h[-2]), int(h[-1])
        base_us = float(p[1].split()[0])
        d = p[2].split()
        disp_us, kern = float(d[0]), d[2]
        a_pl, w_pl, terms = {"bf16": (1, 1, 1), "fp16x3": (2, 2, 3), "fp16x2": (1, 2, 2)}[mode]
        out_pl = 1 if (mode == "bf16" or (mode == "fp16x2" and epi == 1)) else 2
        out_b = 2 * out_pl if epi < 2 else 4
        byt = M * K * 2 * a_pl + N * K * 2 * w_pl + M * N * out_b + (M * N * 4 if epi == 2 else 0) + N * 4
        fl = 2.0 * M * N * K
        rows[name] = dict(name=name, epi=epi, M=M, N=N, K=K, base=base_us, disp=disp_us, kern=kern, byt=byt, fl=fl, terms=terms,
                          t_mfma=terms * fl / (PEAK_TF * 1e12) * 1e6, t_hbm=byt / HBM_PEAK * 1e6, t_hbm_ach=byt / HBM_ACH * 1e6)
    return mode, rows


def tagged(name, tags):
    cls, _, st = name.partition(" s")
    return cls in tags or f"{cls}.s{st}" in tags


def main(argv):
    mode, rows = parse(argv[1])
    title = mode
    if len(argv) > 3:
        mode2, rows2 = parse(argv[2])
        assert mode == "fp16x3" and mode2 == "fp16x2", (mode, mode2)
        tags = set(argv[3].split(","))
        rows = {n: (rows2[n] if tagged(n, tags) and n in rows2 else r) for n, r in rows.items()}
        title = f"fp16x3m (two terms in {argv[3]})"
    print(f"operand mode: {title}")
    print()
    print("| shape | epi | M | N | K | terms | 128-tile us | dispatched us | kernel | TFLOP/s | alg. MB | MFMA floor us | HBM floor us "
          "(8 TB/s / 6.3 TB/s) | binds | frac of binding floor |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    tot_t = tot_floor = tot_fl = tot_ex = 0.0
    for r in rows.values():
        bound = "mfma" if r["t_mfma"] >= r["t_hbm"] else "hbm"
        floor = max(r["t_mfma"], r["t_hbm"])
        print(f"| {r['name']} | {r['epi']} | {r['M']} | {r['N']} | {r['K']} | {r['terms']} | {r['base']:.1f} | {r['disp']:.1f} | {r['kern']} | "
              f"{r['fl'] / r['disp'] / 1e6:.0f} | {r['byt'] / 1e6:.0f} | {r['t_mfma']:.1f} | {r['t_hbm']:.1f} / {r['t_hbm_ach']:.1f} | {bound} | "
              f"{floor / r['disp']:.2f} |")
        n = r["name"]
        if n[-2:-1] == "s":
            w = LAYERS[int(n[-1])] if not n.startswith("merge") else 1
            tot_t += w * r["disp"]; tot_floor += w * floor; tot_fl += w * r["fl"]; tot_ex += w * r["fl"] * r["terms"]
    print()
    print(f"Whole encoder (layer counts 2/2/18/2, one merge per stage boundary): {tot_t / 1e3:.2f} ms of GEMM per group, "
          f"{tot_fl / tot_t / 1e6:.0f} TFLOP/s (algorithmic) average = {tot_fl / tot_t / 1e6 / PEAK_TF:.3f} of the 16-bit MFMA peak"
          f" ({tot_ex / tot_t / 1e6 / PEAK_TF:.3f} counting the executed terms, {tot_ex / tot_fl:.3f} per product on average)"
          f"; sum of the per-shape binding floors {tot_floor / 1e3:.2f} ms = {tot_floor / tot_t:.2f} of the measured time.")


if __name__ == "__main__":
    main(sys.argv)
