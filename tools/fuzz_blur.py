#!/usr/bin/env python3
"""Randomised bit-exactness sweep of the fused blur alone (round 6: the staggered form has more LDS hand-offs than the tests' eight shapes
exercise): random shapes (rows of whole 16-byte vectors, up to ~2^22 voxels), every filter the pyramid uses, level + DoG / level only / DoG
only / level + DoG + half-size volume, z chunks, tile, rows per thread and SIFT3D_TUNE_FUSED_STAGGER drawn at random; every output compared
with the oracle bit for bit.  usage: python tools/fuzz_blur.py [n_cases] [seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _oracle   # test infrastructure (this is a development check, not product code)
pkg = importlib.import_module("3d_sift_cuda_amd")
orc = _oracle.load()
SIG = [1.2262736558914185, 1.5198684930801392, 1.5450079441070557, 1.9465880393981934, 2.452547311782837, 3.0900158882141113, 0.95, 0.5]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    bad = 0
    for i in range(n_cases):
        nx = 4 * int(rng.integers(2, 80)); ny = int(rng.integers(2, 150)); nz = int(rng.integers(2, 150))
        while nx * ny * nz > (1 << 22):
            nz = max(2, nz // 2)
        vol = (rng.standard_normal((nz, ny, nx)) * 40).astype(np.float32)
        if rng.random() < 0.3:
            vol[rng.random(vol.shape) < 0.5] = 0.0
        chunks = int(rng.choice([0, 0, 1, 2, 3, 5])); tile = int(rng.choice([0, 1, 2])); rows = int(rng.choice([0, 2, 2, 1])); stg = int(rng.choice([0, 1, 2, 2]))
        with pkg.Context(nx, ny, nz) as ctx:
            ctx.set_tuning(pkg.TUNE_BLUR_FUSED, 2)
            ctx.set_tuning(pkg.TUNE_FUSED_CHUNKS, chunks); ctx.set_tuning(pkg.TUNE_FUSED_TILE, tile)
            ctx.set_tuning(pkg.TUNE_FUSED_ROWS, rows); ctx.set_tuning(pkg.TUNE_FUSED_STAGGER, stg)
            d_in = torch.from_numpy(vol).cuda(); d_out = torch.empty_like(d_in); d_dog = torch.empty_like(d_in)
            d_half = torch.empty((max(1, nz // 2), max(1, ny // 2), nx // 2), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            for s in rng.choice(SIG, 3, replace=False):
                want = orc.blur(vol, float(s)); wdog = orc.dog(vol, want)
                kind = int(rng.integers(0, 4))
                d_out.fill_(-3.0); d_dog.fill_(-3.0); torch.cuda.synchronize()
                if kind == 0:
                    ctx.gauss_blur_dog_dev(d_in.data_ptr(), d_out.data_ptr(), d_dog.data_ptr(), nx, ny, nz, float(s)); ctx.sync()
                    ok = (bits(d_out.cpu().numpy()) == bits(want)).all() and (bits(d_dog.cpu().numpy()) == bits(wdog)).all()
                elif kind == 1:
                    ctx.gauss_blur_dev(d_in.data_ptr(), d_out.data_ptr(), nx, ny, nz, float(s)); ctx.sync()
                    ok = (bits(d_out.cpu().numpy()) == bits(want)).all()
                elif kind == 2:
                    ctx.gauss_blur_dog_dev(d_in.data_ptr(), 0, d_dog.data_ptr(), nx, ny, nz, float(s)); ctx.sync()
                    ok = (bits(d_dog.cpu().numpy()) == bits(wdog)).all()
                else:
                    if ny < 2 or nz < 2:
                        continue
                    ctx.gauss_blur_dog_half_dev(d_in.data_ptr(), d_out.data_ptr(), d_dog.data_ptr(), d_half.data_ptr(), nx, ny, nz, float(s)); ctx.sync()
                    half = orc.subsample(want)
                    ok = (bits(d_out.cpu().numpy()) == bits(want)).all() and (bits(d_dog.cpu().numpy()) == bits(wdog)).all() and \
                        (bits(d_half.cpu().numpy()[:half.shape[0], :half.shape[1], :half.shape[2]]) == bits(half)).all()
                if not ok:
                    bad += 1
                    print("MISMATCH case %d: dims %s sigma %r kind %d chunks %d tile %d rows %d stagger %d" % (i, (nx, ny, nz), float(s), kind, chunks, tile, rows, stg), flush=True)
        if (i + 1) % 20 == 0:
            print("%d cases, %d mismatches" % (i + 1, bad), flush=True)
    print("fuzz_blur: %d cases, %d mismatches" % (n_cases, bad))
    sys.exit(1 if bad else 0)


main()
