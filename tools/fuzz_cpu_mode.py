#!/usr/bin/env python3
"""Random geometries through the product's Caffe::CPU mode (escoin_forward_cpu, float and double) against the oracle:
bit-equal or it is a failure.  No GPU needed.     python tools/fuzz_cpu_mode.py [cases] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pkg, oracle = ge.load_package(), ge.load_oracle()
    if len(sys.argv) > 3:
        pkg.cpu_kernel_select(sys.argv[3])                         # "avx2" | "avx512" | "auto"
    rng = np.random.RandomState(seed)
    done = bad = 0
    while done < cases:
        grp = int(rng.choice([1, 1, 2, 3, 4]))
        Cg, Mg = int(rng.randint(1, 20)), int(rng.randint(1, 20))
        KH, KW = int(rng.randint(1, 8)), int(rng.randint(1, 8))
        sh, sw = int(rng.choice([1, 1, 1, 2, 3])), int(rng.choice([1, 1, 1, 2, 3]))
        dh, dw = int(rng.choice([1, 1, 2, 3])), int(rng.choice([1, 1, 2, 3]))
        ph, pw = int(rng.randint(0, 6)), int(rng.randint(0, 6))
        H, W = int(rng.randint(1, 40)), int(rng.randint(1, 70))
        if (H + 2 * ph - (dh * (KH - 1) + 1)) < 0 or (W + 2 * pw - (dw * (KW - 1) + 1)) < 0:
            continue
        N = int(rng.randint(1, 12))
        C_, M = Cg * grp, Mg * grp
        dtype = np.float64 if rng.rand() < 0.4 else np.float32
        dens = float(rng.choice([0.02, 0.1, 0.3, 0.7, 1.0]))
        x = rng.uniform(-1, 1, (N, C_, H, W)).astype(dtype)
        w = (rng.uniform(-1, 1, (M, Cg, KH, KW)) * (rng.uniform(size=(M, Cg, KH, KW)) < dens)).astype(dtype)
        b = rng.uniform(-0.1, 0.1, M).astype(dtype) if rng.rand() < 0.7 else None
        relu = bool(rng.rand() < 0.3)
        nt = int(rng.choice([1, 2, 5, 8, 16]))
        g = oracle.geom(C_, H, W, M, KH, KW, ph, pw, sh, sw, dh, dw, grp)
        fwd = oracle.conv_forward_f64 if dtype == np.float64 else oracle.conv_forward
        want = fwd(g, x, w, b, relu=relu, gate=False)
        plan = pkg.Plan(pkg.ConvDesc(N, C_, H, W, M, KH, KW, ph, pw, sh, sw, dh, dw, grp, int(b is not None), int(relu)))
        plan.weight_align_cpu(w)
        cb = int(rng.choice([0, 0, 1, 2, 3, 5]))                  # channels per block forced on half the cases (round 6)
        if cb:
            plan.set_option("cpu_channel_block", cb)
        plan.set_option("cpu_images_per_job", int(rng.choice([0, 0, 1, 2, 3])))   # small images: one, two or three to a job
        got = plan.forward_cpu(x, b, n_threads=nt)
        if not np.array_equal(got, want):
            print("(cpu_channel_block = %d)" % cb)
            bad += 1
            print("MISMATCH", dict(N=N, C=C_, H=H, W=W, M=M, KH=KH, KW=KW, ph=ph, pw=pw, sh=sh, sw=sw, dh=dh, dw=dw, grp=grp,
                                   dtype=dtype.__name__, relu=relu, bias=b is not None, threads=nt), flush=True)
        plan.close()
        done += 1
    print("fuzz_cpu_mode: %d geometries, seed %d, %d mismatches (%s)" % (done, seed, bad, pkg.cpu_kernel_name()))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
