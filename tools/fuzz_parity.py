#!/usr/bin/env python3
"""Randomised differential run: HIP path vs the CPU oracle on random shapes, parameters and data
(development tool, run on the GPU box; the committed test-suite holds the fixed cases).
Usage: python3 tools/fuzz_parity.py [cases] [seed] | batch [cases] [seed] | rtl [cases] [seed] | tol [cases] [seed] | stream [cases] [seed]"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python"))
sys.path.insert(0, str(ROOT / "oracle"))
os.environ.setdefault("OFLK_QUIET", "1")

import lucas_kanade_core as K  # noqa: E402
import lucas_kanade_pyramidal as P  # noqa: E402
import oflk_oracle as O  # noqa: E402


def same(a, b):
    return np.array_equal(a, b)   # -0.0 == +0.0, NaN never produced from finite inputs


def batch_main(cases, seed):
    """large launches (many small pairs per call): the chained-tile / strip-per-XCD path"""
    import ctypes

    import _oflk

    rng = np.random.default_rng(seed)
    f32p = ctypes.POINTER(ctypes.c_float)
    bad = 0
    for i in range(cases):
        H, W = int(rng.integers(30, 260)), int(rng.integers(30, 400))
        B = int(rng.integers(40, 520))
        L, K = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        win = int(rng.choice([3, 5, 5, 5, 7]))
        distinct = []
        for _ in range(3):
            a = rng.normal(110, 45, (H, W)).astype(np.float32)
            distinct.append((a, (a + rng.normal(0, 6, (H, W))).astype(np.float32)))
        distinct.append((distinct[0][0], distinct[0][0].copy()))
        expect = [O.lucas_kanade_pyramidal_ex(a, b, L, win, K) for a, b in distinct]
        order = rng.integers(0, 4, B)
        prev = np.stack([distinct[k][0] for k in order])
        curr = np.stack([distinct[k][1] for k in order])
        u, v = np.empty_like(prev), np.empty_like(prev)
        runs = np.zeros((B, L), np.int32)
        _oflk.check(_oflk.lib().oflk_pyramidal_batch(prev.ctypes.data_as(f32p), curr.ctypes.data_as(f32p), B, H, W, L, win,
                                                     K, u.ctypes.data_as(f32p), v.ctypes.data_as(f32p), None,
                                                     runs.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        ok = all(same(u[j], expect[k][0]) and same(v[j], expect[k][1]) and list(runs[j]) == list(expect[k][3])
                 for j, k in enumerate(order))
        if not ok:
            bad += 1
            print(f"MISMATCH batch case {i}: B={B} H={H} W={W} L={L} K={K} win={win}", flush=True)
        if i % 10 == 9:
            print(f"{i + 1} batch cases, {bad} mismatches", flush=True)
    print(f"done: {cases} batch cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


def rtl_main(cases, seed):
    """RTL-bit-accurate integer mode: GPU per-element flows vs the closed form of the RTL (oracle/rtl_model.py), and --
    on the small cases -- the sampled vector sequence vs the cycle-by-cycle execution (oracle/rtl_cycle_sim.py)"""
    import rtl_cycle_sim as S
    import rtl_golden_model as G
    import rtl_model as M

    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(cases):
        H, W = int(rng.integers(5, 513)), int(rng.integers(5, 1025))
        if i % 4 == 0:
            H, W = int(rng.integers(5, 40)), int(rng.integers(5, 48))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            f0, f1 = rng.integers(0, 256, (H, W)), rng.integers(0, 256, (H, W))
        elif kind == 1:
            f0 = rng.integers(0, 256, (H, W)); f1 = np.roll(f0, (int(rng.integers(-2, 3)), int(rng.integers(-3, 4))), (0, 1))
        elif kind == 2:
            f0 = rng.integers(120, 136, (H, W)); f1 = f0 + rng.integers(-1, 2, (H, W))
        else:
            yy, xx = np.mgrid[0:H, 0:W]
            f0 = (128 + 120 * np.sin(xx / rng.uniform(1.5, 9)) * np.cos(yy / rng.uniform(1.5, 9))).astype(np.int64)
            f1 = np.roll(f0, (1, 2), (0, 1))
        f0, f1 = np.clip(f0, 0, 255).astype(np.uint8), np.clip(f1, 0, 255).astype(np.uint8)
        st = G.rtl_flow_states(f0, f1)
        gx, gy, gt, _ = M.gradient_stream(f0, f1)
        valid, x, y, u, v = M.flow_states(gx, gy, gt, W)
        ok = same(st["u"].astype(np.int64), u) and same(st["v"].astype(np.int64), v) and same(st["valid"], valid) and same(st["x"], x) and same(st["y"], y)
        if ok and H * W <= 2000:
            sim = np.array(S.simulate(f0.reshape(-1).astype(np.int64), f1.reshape(-1).astype(np.int64), W, H), np.int64).reshape(-1, 4)
            vec = G.testbench_vectors(f0, f1)
            ok = vec.shape == sim.shape and same(vec, sim)
        if not ok:
            bad += 1
            print(f"MISMATCH rtl case {i}: H={H} W={W} kind={kind}", flush=True)
        if i % 50 == 49:
            print(f"{i + 1} rtl cases, {bad} mismatches", flush=True)
    print(f"done: {cases} rtl cases, {bad} mismatches")


def tol_main(cases, seed):
    """opt-in tolerant arithmetic: a plan in OFLK_ARITH_TOLERANT against its CPU statement (oracle/oflk_tolerant_model.c),
    bit for bit -- flows and iteration counts -- on random shapes, levels, iteration counts, batch sizes, float32 and uint8"""
    import torch

    import _oflk
    import oflk_tolerant_model as M

    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    bad = 0
    for i in range(cases):
        H, W = int(rng.integers(6, 420)), int(rng.integers(6, 560))
        L, K = int(rng.integers(1, 5)), int(rng.integers(1, 4))
        while L > 1 and (int(H * 0.5 ** (L - 1)) < 1 or int(W * 0.5 ** (L - 1)) < 1):
            L -= 1
        B = int(rng.integers(1, 4))
        kind = int(rng.integers(0, 3))
        prs = []
        for _ in range(B):
            if kind == 0:
                a = rng.integers(0, 256, (H, W)).astype(np.float32)
                b = np.clip(np.roll(a, (int(rng.integers(-2, 3)), int(rng.integers(-3, 4))), (0, 1)) + rng.integers(-5, 6, (H, W)), 0, 255).astype(np.float32)
            elif kind == 1:
                yy, xx = np.mgrid[0:H, 0:W]
                a = np.round(128 + 60 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + rng.normal(0, 4, (H, W))).clip(0, 255).astype(np.float32)
                b = np.round(128 + 60 * np.sin((xx - 1.7) / 7.0) * np.cos((yy + 0.6) / 5.0) + rng.normal(0, 4, (H, W))).clip(0, 255).astype(np.float32)
            else:
                a = (rng.random((H, W)) * 255).astype(np.float32)
                b = (np.roll(a, 1, 1) * np.float32(0.98) + rng.normal(0, 1, (H, W))).astype(np.float32)
            prs.append((a, b))
        prev, curr = np.stack([p for p, _ in prs]), np.stack([c for _, c in prs])
        plan = _oflk.Plan(0, B, H, W, L, 5, K)
        plan.set_arithmetic(2)
        tp, tc = torch.from_numpy(prev).to(dev), torch.from_numpy(curr).to(dev)
        u, v = torch.empty_like(tp), torch.empty_like(tp)
        plan.pyramidal(tp.data_ptr(), tc.data_ptr(), u.data_ptr(), v.data_ptr(), st)
        _, runs = plan.read_log(st)
        torch.cuda.synchronize()
        hu, hv = u.cpu().numpy(), v.cpu().numpy()
        ok = True
        for b_, (a, b) in enumerate(prs):
            mu, mv, _, mruns = M.pyramidal(a, b, M.tolerant_spec(L, K, (H, W)), 5)
            ok &= same(hu[b_], mu) and same(hv[b_], mv) and list(runs[b_]) == list(mruns)
        if kind < 2:
            t8p, t8c = tp.to(torch.uint8), tc.to(torch.uint8)
            plan.pyramidal_u8(t8p.data_ptr(), t8c.data_ptr(), u.data_ptr(), v.data_ptr(), st)
            torch.cuda.synchronize()
            ok &= same(u.cpu().numpy(), hu) and same(v.cpu().numpy(), hv)
        plan.close()
        if not ok:
            bad += 1
            print(f"MISMATCH tol case {i}: B={B} H={H} W={W} L={L} K={K} kind={kind}", flush=True)
        if i % 50 == 49:
            print(f"{i + 1} tol cases, {bad} mismatches", flush=True)
    print(f"done: {cases} tol cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


def stream_main(cases, seed):
    """single-scale 5x5 with the streaming kernel FORCED (oflk_plan_set_kernels(OFLK_KERNELS_STREAM)): exact on 8-bit frames,
    doubtful tiles redone by the tile kernel -- against the oracle, value for value, on every kind of frame"""
    import torch

    import _oflk

    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    bad = 0
    for i in range(cases):
        H, W = int(rng.integers(5, 600)), int(rng.integers(5, 800))
        B = int(rng.integers(1, 4))
        kind = int(rng.integers(0, 5))
        prs = []
        for _ in range(B):
            if kind == 0:      # 8-bit noise
                a = rng.integers(0, 256, (H, W)).astype(np.float32)
                b = np.roll(a, int(rng.integers(-3, 4)), axis=1)
            elif kind == 1:    # 8-bit texture with hard 0 / 255 patches (windows over the bound)
                yy, xx = np.mgrid[0:H, 0:W]
                a = np.round(128 + 60 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + rng.normal(0, 4, (H, W))).clip(0, 255)
                for _ in range(4):
                    y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
                    a[y:y + 9, x:x + 13] = 255.0 * ((np.add.outer(np.arange(min(9, H - y)), np.arange(min(13, W - x))) // 2) % 2)
                a = a.astype(np.float32)
                b = np.roll(a, (1, 2), (0, 1))
            elif kind == 2:    # not integers
                a = (rng.random((H, W)) * 255).astype(np.float32)
                b = (np.roll(a, 1, 1) * np.float32(0.99)).astype(np.float32)
            elif kind == 3:    # signed, wide dynamic range
                a = (rng.normal(0, 1, (H, W)) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
                b = (a + rng.normal(0, 0.1, (H, W)) * 10.0 ** rng.integers(-3, 3)).astype(np.float32)
            else:              # 8-bit with a few stray pixels (fractional, negative, above 255)
                a = rng.integers(0, 256, (H, W)).astype(np.float32)
                b = np.roll(a, (1, -1), (0, 1)).copy()
                for _ in range(3):
                    a[int(rng.integers(0, H)), int(rng.integers(0, W))] = np.float32(rng.choice([0.5, -3.0, 255.25, 1000.0]))
                    b[int(rng.integers(0, H)), int(rng.integers(0, W))] = np.float32(rng.choice([17.75, -0.5, 256.0]))
            prs.append((a, b))
        prev, curr = np.stack([p for p, _ in prs]), np.stack([c for _, c in prs])
        win = int(rng.choice([4, 5, 5, 6, 7, 7]))
        plan = _oflk.Plan(0, B, H, W, 1, win, 0)
        plan.set_kernels(2)
        tp, tc = torch.from_numpy(prev).to(dev), torch.from_numpy(curr).to(dev)
        u, v = torch.empty_like(tp), torch.empty_like(tp)
        plan.single_scale(tp.data_ptr(), tc.data_ptr(), u.data_ptr(), v.data_ptr(), st)
        torch.cuda.synchronize()
        hu, hv = u.cpu().numpy(), v.cpu().numpy()
        ok = True
        for b_, (a, b) in enumerate(prs):
            ou, ov = O.lucas_kanade_single_scale(a, b, win)
            ok &= same(hu[b_], ou) and same(hv[b_], ov)
        if kind in (0, 1):
            t8p, t8c = tp.to(torch.uint8), tc.to(torch.uint8)   # (kept alive until the kernels have run)
            plan.single_scale_u8(t8p.data_ptr(), t8c.data_ptr(), u.data_ptr(), v.data_ptr(), st)
            torch.cuda.synchronize()
            ok &= same(u.cpu().numpy(), hu) and same(v.cpu().numpy(), hv)
        plan.close()
        if not ok:
            bad += 1
            print(f"MISMATCH stream case {i}: B={B} H={H} W={W} kind={kind} win={win}", flush=True)
        if i % 50 == 49:
            print(f"{i + 1} stream cases, {bad} mismatches", flush=True)
    print(f"done: {cases} stream cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "stream":
        return stream_main(int(sys.argv[2]) if len(sys.argv) > 2 else 200, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    if len(sys.argv) > 1 and sys.argv[1] == "tol":
        return tol_main(int(sys.argv[2]) if len(sys.argv) > 2 else 200, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    if len(sys.argv) > 1 and sys.argv[1] == "rtl":
        return rtl_main(int(sys.argv[2]) if len(sys.argv) > 2 else 200, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    if len(sys.argv) > 1 and sys.argv[1] == "batch":
        return batch_main(int(sys.argv[2]) if len(sys.argv) > 2 else 30, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(cases):
        H = int(rng.integers(1, int(os.environ.get("FUZZ_MAXH", 220))))
        W = int(rng.integers(1, int(os.environ.get("FUZZ_MAXW", 300))))
        win = int(rng.choice([3, 4, 5, 5, 5, 7, 9, 11]))
        if i % 10 == 9:    # the generic-window kernel (1x1, 13x13 ...): exact and slow, so small frames
            win = int(rng.choice([1, 12, 13, 15, 19, 27]))
            H, W = min(H, 90), min(W, 110)
        kind = int(rng.integers(0, 4))
        if kind == 0:      # 8-bit frames
            a = rng.integers(0, 256, (H, W)).astype(np.float32)
            b = np.roll(a, int(rng.integers(-3, 4)), axis=1)
        elif kind == 1:    # smooth texture + noise
            yy, xx = np.mgrid[0:H, 0:W]
            a = (128 + 60 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + rng.normal(0, 4, (H, W))).astype(np.float32)
            b = (128 + 60 * np.sin((xx - 1.7) / 7.0) * np.cos((yy + 0.6) / 5.0) + rng.normal(0, 4, (H, W))).astype(np.float32)
        elif kind == 2:    # signed, wide dynamic range
            a = (rng.normal(0, 1, (H, W)) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
            b = (a + rng.normal(0, 0.1, (H, W)) * 10.0 ** rng.integers(-3, 3)).astype(np.float32)
        else:              # mostly flat with a few features
            a = np.full((H, W), 77.0, np.float32)
            for _ in range(5):
                y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
                a[max(y - 3, 0):y + 4, max(x - 3, 0):x + 4] += rng.normal(0, 30, a[max(y - 3, 0):y + 4, max(x - 3, 0):x + 4].shape).astype(np.float32)
            b = np.roll(a, 1, axis=0)
        ok = True
        u, v = K.lucas_kanade_single_scale(a, b, win)
        ou, ov = O.lucas_kanade_single_scale(a, b, win)
        ok &= same(u, ou) and same(v, ov)
        L = int(rng.integers(1, 5))
        it = int(rng.integers(0, 4))
        while L > 1 and (int(H * 0.5 ** (L - 1)) < 1 or int(W * 0.5 ** (L - 1)) < 1):
            L -= 1
        gu, gv, glog, gruns = P.lucas_kanade_pyramidal_with_log(a, b, L, win, it)
        eu, ev, elog, eruns = O.lucas_kanade_pyramidal_ex(a, b, L, win, it)
        ok &= list(gruns[:L]) == list(eruns) and same(gu, eu) and same(gv, ev)
        if kind == 0:      # the same 8-bit frames as uint8 arrays: the kernels read the bytes themselves
            a8, b8 = a.astype(np.uint8), b.astype(np.uint8)
            u8, v8 = K.lucas_kanade_single_scale(a8, b8, win)
            ok &= same(u8, ou) and same(v8, ov)
            hu, hv, _, hruns = P.lucas_kanade_pyramidal_with_log(a8, b8, L, win, it)
            ok &= list(hruns[:L]) == list(eruns) and same(hu, eu) and same(hv, ev)
        if not ok:
            bad += 1
            print(f"MISMATCH case {i}: H={H} W={W} win={win} kind={kind} L={L} iters={it} runs gpu={list(gruns[:L])} oracle={list(eruns)}", flush=True)
        if i % 50 == 49:
            print(f"{i + 1} cases, {bad} mismatches", flush=True)
    print(f"done: {cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
