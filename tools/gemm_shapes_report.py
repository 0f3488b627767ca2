#!/usr/bin/env python3
"""Per-shape roofline table from tools/gemm_lab output: for every encoder GEMM shape the algorithmic HBM bytes, the two
floors (16-bit MFMA dense peak 2.5 PFLOP/s; HBM 8 TB/s peak and the 6.3 TB/s the guide measures as achievable), which
one binds, and the measured fraction of the binding floor. A table of the split-operand mode (first line says fp16x3):
16-bit operands and outputs are two planes (twice the bytes), the matrix pipe executes three terms per algorithmic product
(MFMA floor = 3 x 2MNK / peak), TFLOP/s stay algorithmic.   usage: gemm_shapes_report.py lab.txt"""
import sys

PEAK_TF, HBM_PEAK, HBM_ACH = 2500.0, 8.0e12, 6.29e12
LAYERS = {0: 2, 1: 2, 2: 18, 3: 2}


def main(path):
    rows = []
    split = "fp16x3" in open(path).readline()
    pl, terms = (2, 3) if split else (1, 1)
    for ln in open(path):
        p = ln.split("|")
        if len(p) < 4 or p[0].startswith("shape"):
            continue
        h = p[0].split()
        name, epi, M, N, K = " ".join(h[:-4]), int(h[-4]), int(h[-3]), int(h[-2]), int(h[-1])
        base_us = float(p[1].split()[0])
        d = p[2].split()
        disp_us, kern = float(d[0]), d[2]
        out_b = 2 * pl if epi < 2 else 4
        byt = M * K * 2 * pl + N * K * 2 * pl + M * N * out_b + (M * N * 4 if epi == 2 else 0) + N * 4
        fl = 2.0 * M * N * K
        t_mfma = terms * fl / (PEAK_TF * 1e12) * 1e6
        t_hbm = byt / HBM_PEAK * 1e6
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        floor = max(t_mfma, t_hbm)
        rows.append((name, epi, M, N, K, base_us, disp_us, kern, byt, t_mfma, t_hbm, byt / HBM_ACH * 1e6, bound, floor))
    print("| shape | epi | M | N | K | 128-tile us | dispatched us | kernel | TFLOP/s | alg. MB | MFMA floor us | HBM floor us "
          "(8 TB/s / 6.3 TB/s) | binds | frac of binding floor |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    tot_t = tot_floor = tot_fl = 0.0
    for (name, epi, M, N, K, bu, du, kern, byt, tm, th, tha, bound, floor) in rows:
        print(f"| {name} | {epi} | {M} | {N} | {K} | {bu:.1f} | {du:.1f} | {kern} | {2.0 * M * N * K / du / 1e6:.0f} | "
              f"{byt / 1e6:.0f} | {tm:.1f} | {th:.1f} / {tha:.1f} | {bound} | {floor / du:.2f} |")
        if name[-2:-1] == "s":
            w = LAYERS[int(name[-1])] if not name.startswith("merge") else 1
            tot_t += w * du; tot_floor += w * floor; tot_fl += w * 2.0 * M * N * K
    print()
    print(f"Whole encoder (layer counts 2/2/18/2, one merge per stage boundary): {tot_t / 1e3:.2f} ms of GEMM per group, "
          f"{tot_fl / tot_t / 1e6:.0f} TFLOP/s (algorithmic) average = {tot_fl / tot_t / 1e6 / PEAK_TF:.3f} of the 16-bit MFMA peak"
          + (f" ({terms * tot_fl / tot_t / 1e6 / PEAK_TF:.3f} counting the {terms} executed terms)" if terms > 1 else "")
          + f"; sum of the per-shape binding floors {tot_floor / 1e3:.2f} ms = {tot_floor / tot_t:.2f} of the measured time.")


if __name__ == "__main__":
    main(sys.argv[1])
