#!/usr/bin/env python3
"""tools/mfma_numerics.py -- which arithmetic does ONE gfx950 matrix instruction perform on its K products and C?

The L2 pre-filters (mfma_argmin_kernel, mfma_scan_kernel, mfma_dense_kernel) decide rows by |x|^2 - 2 q.x computed on the
matrix cores and need a DETERMINISTIC bound of its rounding error.  Charging one unit roundoff per product (what a chain of
scalar fmaf would cost) is safe but 5-10 x wider than what happens; this script finds out what happens: it runs
build/tools/mfma_numerics (one instruction per case: all outputs = sum_k a_k b_k + c) on targeted and random cases and
compares the hardware's bits with exact rational models:

  exact        RNE_fp32(c + sum of all K exact products)              -- one rounding per instruction
  seq          s = c; s = RNE(s + p_k) for k = 0 .. K-1                -- an fmaf chain: K roundings
  blockB       s = c; s = RNE(s + exact sum of B consecutive products) -- K / B roundings
  *_rz         the same with truncation instead of round-to-nearest-even
  alignG       every term aligned to the largest exponent and truncated to 24 + G bits before an exact sum, then RNE

usage (on the GPU box):  python tools/mfma_numerics.py [cases_per_shape] > report
"""
import os
import subprocess
import sys
import tempfile
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "tools", "mfma_numerics")
SHAPES = {"h32": (16, np.float16), "h16": (32, np.float16), "s32": (2, np.float32), "s16": (4, np.float32)}


def rnd_f32(fr, mode="rne"):
    """a Fraction rounded to fp32 (as a Fraction): round-to-nearest-even or toward zero; overflow is not expected here"""
    if fr == 0:
        return Fraction(0)
    sign = -1 if fr < 0 else 1
    m = abs(fr)
    e = m.numerator.bit_length() - m.denominator.bit_length()
    if Fraction(2) ** e > m:
        e -= 1
    e = max(e, -126)
    ulp = Fraction(2) ** (e - 23)
    q = m / ulp
    n = q.numerator // q.denominator
    rem = q - n
    if mode == "rne" and (rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1)):
        n += 1
    return sign * n * ulp


def model_block(p, c, B, mode="rne"):
    s = c
    for i in range(0, len(p), B):
        s = rnd_f32(s + sum(p[i:i + B], Fraction(0)), mode)
    return s


def model_align(p, c, G):
    terms = [c] + list(p)
    nz = [abs(t) for t in terms if t != 0]
    if not nz:
        return Fraction(0)
    m = max(nz)
    e = m.numerator.bit_length() - m.denominator.bit_length()
    if Fraction(2) ** e > m:
        e -= 1
    q = Fraction(2) ** (e - 23 - G)
    tot = Fraction(0)
    for t in terms:
        n = abs(t) / q
        tot += (1 if t >= 0 else -1) * (n.numerator // n.denominator) * q
    return rnd_f32(tot)


def _floor_exp(m):
    e = m.numerator.bit_length() - m.denominator.bit_length()
    if Fraction(2) ** e > m:
        e -= 1
    return e


def model_block_align(p, c, B, G, c_inside, term_mode="rz"):
    """blocks of B products; inside a block every term is aligned to the block's largest exponent and cut to 24 + G bits
    (toward zero, or to nearest), the aligned terms are added exactly, the block's result is RNE'd to fp32.  c_inside:
    the running value is one of the aligned terms; otherwise the products are aligned among themselves and their sum is
    added to the running value exactly before the rounding."""
    s = c
    for i in range(0, len(p), B):
        terms = list(p[i:i + B]) + ([s] if c_inside else [])
        nz = [abs(t) for t in terms if t != 0]
        if not nz:
            continue
        q = Fraction(2) ** (_floor_exp(max(nz)) - 23 - G)
        tot = Fraction(0)
        for t in terms:
            n = abs(t) / q
            k = n.numerator // n.denominator
            if term_mode == "rne":
                rem = n - k
                if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and k % 2 == 1):
                    k += 1
            tot += (1 if t >= 0 else -1) * k * q
        s = rnd_f32(tot if c_inside else s + tot)
    return s


def models(K):
    ms = {"exact": lambda p, c: model_block(p, c, K), "seq": lambda p, c: model_block(p, c, 1),
          "exact_rz": lambda p, c: model_block(p, c, K, "rz"), "seq_rz": lambda p, c: model_block(p, c, 1, "rz")}
    for B in (2, 4, 8, 16):
        if B < K:
            ms["block%d" % B] = (lambda B: lambda p, c: model_block(p, c, B))(B)
            ms["block%d_rz" % B] = (lambda B: lambda p, c: model_block(p, c, B, "rz"))(B)
    for G in (0, 1, 2, 3, 4, 8):
        ms["align%d" % G] = (lambda G: lambda p, c: model_align(p, c, G))(G)
    if K >= 16:
        for G in range(0, 8):
            ms["b8a%d_out" % G] = (lambda G: lambda p, c: model_block_align(p, c, 8, G, False))(G)
            ms["b8a%d_in" % G] = (lambda G: lambda p, c: model_block_align(p, c, 8, G, True))(G)
            ms["b8a%d_in_rne" % G] = (lambda G: lambda p, c: model_block_align(p, c, 8, G, True, "rne"))(G)
    # the products of one half of the K range first, then the other (lanes 0-31 | 32-63), each half exact
    ms["halves"] = lambda p, c: model_block(p, c, max(K // 2, 1))
    return ms


def targeted(K, dt):
    """(name, a, b, c) cases that separate the models"""
    out = []
    one = np.ones(K, dtype=dt)

    def prod_cases(name, prods, c):
        a = np.zeros(K, dtype=dt)
        b = np.zeros(K, dtype=dt)
        for k, pr in enumerate(prods):
            if pr == 0:
                continue
            # p = 2^e split over the two factors so that both are normal in the element type
            e = int(np.round(np.log2(abs(pr))))
            assert abs(pr) == 2.0 ** e
            ea = e // 2
            a[k] = dt(np.sign(pr) * 2.0 ** ea)
            b[k] = dt(2.0 ** (e - ea))
        out.append((name, a, b, np.float32(c)))

    for e in (-24, -25, -26, -27, -28):
        prod_cases("c=1, every product 2^%d" % e, [2.0 ** e] * K, 1.0)
    for k in range(K):
        pr = [0.0] * K
        pr[k] = 2.0 ** -24
        pr[(k + 1) % K] = 2.0 ** -26 if dt == np.float32 else 2.0 ** -26
        prod_cases("c=1, half ulp at k=%d, a sticky 2^-26 after it" % k, pr, 1.0)
    for k in range(K):
        pr = [0.0] * K
        pr[k] = 1.0
        pr[(k + K // 2) % K] = 2.0 ** -28
        prod_cases("c=-1, +1 at k=%d, 2^-28 elsewhere (cancellation)" % k, pr, -1.0)
    prod_cases("c=2^20, products 2^-4 (each below half an ulp of c)", [2.0 ** -4] * K, 2.0 ** 20)
    prod_cases("c=2^20, products 2^-5", [2.0 ** -5] * K, 2.0 ** 20)
    prod_cases("c=0, 1 then -1 then 2^-26s", [1.0, -1.0] + [2.0 ** -26] * (K - 2) if K > 2 else [1.0, -1.0], 0.0)
    return out


def random_cases(K, dt, n, rng):
    cases = []
    for i in range(n):
        mode = i % 6
        if mode == 0:
            a = rng.standard_normal(K)
            b = rng.standard_normal(K)
            c = rng.standard_normal() * 4
        elif mode == 1:
            a = rng.standard_normal(K) * 2.0 ** rng.integers(-6, 7, K)
            b = rng.standard_normal(K) * 2.0 ** rng.integers(-6, 7, K)
            c = rng.standard_normal() * 2.0 ** rng.integers(-8, 9)
        elif mode == 2:   # a long-running accumulator: c much larger than the products
            a = rng.random(K)
            b = rng.random(K)
            c = rng.random() * 2.0 ** rng.integers(4, 12)
        elif mode == 3:   # cancellation
            a = rng.standard_normal(K)
            b = rng.standard_normal(K)
            c = -float(np.dot(a.astype(dt).astype(np.float64), b.astype(dt).astype(np.float64))) * (1 + rng.standard_normal() * 1e-3)
        elif mode == 4:   # ties: everything a multiple of a small power of two
            a = rng.integers(-8, 9, K) * 2.0 ** -6
            b = rng.integers(-8, 9, K) * 2.0 ** -7
            c = float(rng.integers(1, 1 << 12)) * 2.0 ** rng.integers(0, 12)
        else:             # what the kernels see: data in [0, 1), accumulator = partial dot product
            a = rng.random(K)
            b = rng.random(K)
            c = rng.random() * K * rng.integers(1, 200) * 0.25
        cases.append(("random mode %d" % mode, a.astype(dt), b.astype(dt), np.float32(c)))
    return cases


def run_shape(shape, n_random, rng, tmp):
    K, dt = SHAPES[shape]
    cases = targeted(K, dt) + random_cases(K, dt, n_random, rng)
    a = np.stack([c[1] for c in cases]).astype(dt)
    b = np.stack([c[2] for c in cases]).astype(dt)
    c = np.array([c[3] for c in cases], dtype=np.float32)
    path, res = os.path.join(tmp, shape + ".bin"), os.path.join(tmp, shape + ".out")
    with open(path, "wb") as f:
        f.write(a.tobytes())
        f.write(b.tobytes())
        f.write(c.tobytes())
    r = subprocess.run([EXE, shape, path, res], capture_output=True, text=True)
    print(r.stdout.strip(), r.stderr.strip())
    if r.returncode != 0:
        return
    hw = np.fromfile(res, dtype=np.float32)
    keep = os.path.join(ROOT, "gpurun_out", "r06d")
    if os.path.isdir(keep):   # the cases and the hardware's answers, for models written later
        np.savez(os.path.join(keep, "mfma_cases_%s.npz" % shape), a=a, b=b, c=c, hw=hw)
    ms = models(K)
    hits = {m: 0 for m in ms}
    first_miss = {}
    worst = {"seq": Fraction(0), "exact": Fraction(0)}
    for i, (name, av, bv, cv) in enumerate(cases):
        p = [Fraction(float(x)) * Fraction(float(y)) for x, y in zip(av, bv)]
        cf = Fraction(float(cv))
        got = Fraction(float(hw[i]))
        for m, fn in ms.items():
            if fn(p, cf) == got:
                hits[m] += 1
            elif m not in first_miss:
                first_miss[m] = (i, name)
        # the hardware's error against the exact sum, in units of u x (|c| + sum |p|)
        mag = abs(cf) + sum(abs(x) for x in p)
        if mag:
            err = abs(got - (cf + sum(p))) / (mag * Fraction(1, 1 << 24))
            worst["exact"] = max(worst["exact"], err)
    ncase = len(cases)
    print("shape %s  K = %d  %s  cases %d (targeted %d)" % (shape, K, dt.__name__, ncase, ncase - n_random))
    for m in sorted(ms, key=lambda m: -hits[m]):
        print("   %-12s matches %6d / %d%s" % (m, hits[m], ncase, "" if hits[m] == ncase else
                                               "   first miss: case %d (%s)" % first_miss[m]))
    print("   largest |hardware - exact sum| / (u (|c| + sum |p_k|)) over all cases: %.4f   (1 rounding of the result: <= 1; "
          "K sequential roundings: up to K = %d)" % (float(worst["exact"]), K))
    # the targeted cases in full
    for i, (name, av, bv, cv) in enumerate(cases[:ncase - n_random]):
        p = [Fraction(float(x)) * Fraction(float(y)) for x, y in zip(av, bv)]
        cf = Fraction(float(cv))
        print("   T%-3d %-62s hw %-18r exact-RNE %-18r seq %r" % (
            i, name, float(hw[i]), float(model_block(p, cf, K)), float(model_block(p, cf, 1))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    rng = np.random.default_rng(6)
    with tempfile.TemporaryDirectory() as tmp:
        for shape in ("h32", "h16", "s32", "s16"):
            run_shape(shape, n, rng, tmp)
            print()


if __name__ == "__main__":
    main()
