#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle of the RTL-bit-accurate integer mode) -- never on
the product path: only tests/ may import this.

Literal, cycle-by-cycle execution of the reference's single-scale RTL pipeline
(rtl/common/frame_buffer_simple.sv, rtl/common/line_buffer_5x5.sv:48-151; rtl/unopt/gradient_compute.sv:89-159,
window_accumulator.sv:100-189, flow_solver.sv:52-166, optical_flow_top.sv) and of the monitor loop of
tb/tb_optical_flow_top.sv:176-240: what the testbench would sample from `flow_valid / flow_x / flow_y /
flow_u / flow_v`, vector by vector.  Registers update with non-blocking semantics: every process computes
its next state from the current state, then all commit together.

PARITY UNPINNED: Vivado / xsim is not available here and the reference holds no output of the RTL as
committed.  The only golden data, the xsim log in README.md:455-531, was produced by an earlier revision of
the solver and testbench: this simulation reproduces its vector count to within two (73 287 against 73 289)
and its zero samples but not its values (first vector at (3, 2) against (10, 8); region mean u = 0.73
against -0.765).  The reference's other artefact, results/flow_visualization.png (statistics of a run on the
mountain frames: 73 289 vectors, 856 in the test region, mean u = 1.459), is matched in geometry (856) and to 5 % in
value (1.388): tests/test_rtl_model.py.  A later round with a simulator, or a maintainer's fresh log, can pin it.

Usage: python3 oracle/rtl_cycle_sim.py [sinusoid|natural]   (the tb's summary for tests/golden/rtl_frames.npz)
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]


def s8(x):
    x &= 0xFF
    return x - 256 if x >= 128 else x


def sx(x, bits):
    x &= (1 << bits) - 1
    return x - (1 << bits) if x >> (bits - 1) else x


class LineBuf:
    """line_buffer_5x5.sv"""

    def __init__(self, W, H):
        self.W, self.H = W, H
        self.l0 = [0] * W
        self.l1 = [0] * W
        self.l2 = [0] * W
        self.l3 = [0] * W
        self.cur = [0] * 5
        self.col = 0
        self.row = 0
        self.valid = 0

    def window(self):
        c = self.col
        w = []
        for line in (self.l3, self.l2, self.l1, self.l0):
            w.append([line[c - 4] if c >= 4 else 0, line[c - 3] if c >= 3 else 0, line[c - 2] if c >= 2 else 0,
                      line[c - 1] if c >= 1 else 0, line[c]])
        w.append([self.cur[4], self.cur[3], self.cur[2], self.cur[1], self.cur[0]])
        return w

    def coords(self):
        if self.valid and self.col >= 2 and self.row >= 2:
            return self.col - 2, self.row - 2
        return 0, 0

    def step(self, data_in, data_valid):
        """returns a closure that commits the next state"""
        if not data_valid:
            return lambda: None
        c, r = self.col, self.row
        ncur = [data_in] + self.cur[:4]
        if c == self.W - 1:
            ncol = 0
            nrow = 0 if r == self.H - 1 else r + 1
        else:
            ncol, nrow = c + 1, r
        nvalid = 1 if (r >= 4 and c >= 4) else 0
        n3, n2, n1, n0 = self.l2[c], self.l1[c], self.l0[c], self.cur[0]

        def commit():
            self.cur = ncur
            self.col, self.row, self.valid = ncol, nrow, nvalid
            self.l3[c], self.l2[c], self.l1[c], self.l0[c] = n3, n2, n1, n0
        return commit


def trunc_div(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def simulate(f0, f1, W=320, H=240):
    """f0 = previous frame, f1 = current frame, flat uint8 sequences of W * H pixels; returns the list of
    (flow_x, flow_y, flow_u, flow_v) the testbench's monitor loop samples, in order (u, v: S8.7 integers)"""
    TOTAL = W * H
    # ---- state ------------------------------------------------------------------
    fb = dict(cnt=0, streaming=0, valid=0, done=0, curr=0, prev=0)
    lb_c, lb_p = LineBuf(W, H), LineBuf(W, H)
    g = dict(gx=0, gy=0, gt=0, valid=0)
    lb_x, lb_y, lb_t = LineBuf(W, H), LineBuf(W, H), LineBuf(W, H)
    acc = dict(prod=None, valid_d1=0, x_d1=0, y_d1=0, sums=[0] * 5, valid=0, x=0, y=0)
    acc["prod"] = [[0] * 25 for _ in range(5)]
    sol = dict(p=[0] * 6, valid_d1=0, x_d1=0, y_d1=0, u=0, v=0, valid=0, x=0, y=0)
    top = dict(busy=0, done=0)

    outputs = []      # per cycle with flow_valid (as the tb samples them): (x, y, u, v)
    cycle = 0
    start_cycles = {0}   # start is high during the first simulated edge
    monitoring = False
    finished = False
    while not finished:
        start = 1 if cycle in start_cycles else 0
        # ---- tb monitor: samples the registers as they are BEFORE this edge's updates
        if monitoring:
            if sol["valid"]:
                outputs.append((sol["x"], sol["y"], sol["u"], sol["v"]))
        # the tb loop is `while (!done) begin @(posedge clk); sample; end`
        # it starts once busy has been seen; handled below with the same pre-edge view
        # ---- combinational views of the current state ----------------------------
        wc, wp = lb_c.window(), lb_p.window()
        avg = [[(((wc[i + 1][j + 1] + wp[i + 1][j + 1]) & 0x1FF) >> 1) for j in range(3)] for i in range(3)]
        sob_xl = -avg[0][0] - (avg[1][0] << 1) - avg[2][0]
        sob_xr = avg[0][2] + (avg[1][2] << 1) + avg[2][2]
        sob_x = sx((sob_xl + sob_xr) >> 3, 12)
        sob_yt = -avg[0][0] - (avg[0][1] << 1) - avg[0][2]
        sob_yb = avg[2][0] + (avg[2][1] << 1) + avg[2][2]
        sob_y = sx((sob_yt + sob_yb) >> 3, 12)
        temporal = sx((wp[2][2] & 0xFF) - (wc[2][2] & 0xFF), 12)
        g_valid_next = lb_c.valid
        wx, wy, wt = lb_x.window(), lb_y.window(), lb_t.window()
        ax, ay = lb_x.coords()
        # flow solver combinational part on its registered products
        p = sol["p"]
        det = sx(sx(p[0], 32) - sx(p[1], 32), 32)
        nu = sx(sx(p[2], 32) - sx(p[3], 32), 32)
        nv = sx(sx(p[4], 32) - sx(p[5], 32), 32)
        if det > 1000 or det < -1000:
            fu = sx(trunc_div(nu << 7, det), 16)
            fv = sx(trunc_div(nv << 7, det), 16)
            fu = max(-1024, min(1024, fu))
            fv = max(-1024, min(1024, fv))
        else:
            fu = fv = 0
        # ---- next state of every clocked process ---------------------------------
        commits = []
        # frame buffer
        nfb = dict(fb)
        nfb["done"] = 0
        if start and not fb["streaming"]:
            nfb["streaming"], nfb["cnt"], nfb["valid"] = 1, 0, 1
        elif fb["streaming"]:
            if fb["cnt"] == TOTAL - 1:
                nfb["streaming"], nfb["done"], nfb["cnt"] = 0, 1, 0
            else:
                nfb["cnt"] = fb["cnt"] + 1
        else:
            nfb["valid"] = 0
        if fb["streaming"]:
            nfb["curr"], nfb["prev"] = int(f1[fb["cnt"]]), int(f0[fb["cnt"]])
        # gradient line buffers (8-bit data stored as signed)
        commits.append(lb_c.step(s8(fb["curr"]), fb["valid"]))
        commits.append(lb_p.step(s8(fb["prev"]), fb["valid"]))
        ng = dict(gx=sob_x, gy=sob_y, gt=temporal, valid=g_valid_next)
        # accumulator line buffers
        commits.append(lb_x.step(g["gx"], g["valid"]))
        commits.append(lb_y.step(g["gy"], g["valid"]))
        commits.append(lb_t.step(g["gt"], g["valid"]))
        nprod = [[0] * 25 for _ in range(5)]
        for i in range(5):
            for j in range(5):
                a, b, c = wx[i][j], wy[i][j], wt[i][j]
                k = i * 5 + j
                nprod[0][k] = a * a
                nprod[1][k] = b * b
                nprod[2][k] = a * b
                nprod[3][k] = a * c
                nprod[4][k] = b * c
        nacc = dict(prod=nprod, valid_d1=lb_x.valid, x_d1=ax, y_d1=ay,
                    sums=[sx(sum(acc["prod"][q]), 32) for q in range(5)], valid=acc["valid_d1"], x=acc["x_d1"], y=acc["y_d1"])
        sxx, syy, sxy, sxt, syt = acc["sums"]
        nsol = dict(p=[sxx * syy, sxy * sxy, syy * sxt, sxy * syt, sxx * syt, sxy * sxt], valid_d1=acc["valid"],
                    x_d1=acc["x"], y_d1=acc["y"], u=fu, v=fv, valid=sol["valid_d1"], x=sol["x_d1"], y=sol["y_d1"])
        ntop = dict(top)
        if start:
            ntop["busy"], ntop["done"] = 1, 0
        elif fb["done"]:
            ntop["busy"], ntop["done"] = 0, 1
        elif top["done"]:
            ntop["done"] = 0
        # tb control flow (pre-edge view): the monitor loop runs while !done
        if not monitoring and top["busy"]:
            monitoring = True
        if monitoring and top["done"]:
            finished = True
        # ---- commit -----------------------------------------------------------------
        for c in commits:
            c()
        fb, g, acc, sol, top = nfb, ng, nacc, nsol, ntop
        cycle += 1
        if cycle > 3 * TOTAL:
            raise RuntimeError("timeout")
    return outputs


def report(outputs):
    print("Total valid flow vectors:", len(outputs))
    if outputs:
        print("first vector position:", outputs[0][:2])
    px = py = 0
    first = True
    n = 0
    su = sv = squ = sqv = 0.0
    samples = []
    for (x, y, u, v) in outputs:
        if first:
            px, py = x, y
            first = False
        uf, vf = u / 128.0, v / 128.0
        if 55 <= px <= 85 and 105 <= py <= 135:
            su += uf
            sv += vf
            squ += uf * uf
            sqv += vf * vf
            n += 1
            if n % 100 == 1:
                samples.append((px, py, uf, vf))
        px, py = x, y
    print("Vectors in test region:", n)
    if n:
        mu, mv = su / n, sv / n
        print(f"Mean: u={mu:6.3f}, v={mv:6.3f}   Std: u={max(squ / n - mu * mu, 0) ** 0.5:6.3f}, v={max(sqv / n - mv * mv, 0) ** 0.5:6.3f}")
    for smp in samples:
        print("  [x=%3d, y=%3d] u=%6.3f, v=%6.3f" % smp)


def main():
    z = np.load(ROOT / "tests/golden/rtl_frames.npz")
    which = sys.argv[1] if len(sys.argv) > 1 else "sinusoid"
    f0 = z[f"{which}__frame_00"].astype(np.int64)
    f1 = z[f"{which}__frame_01"].astype(np.int64)
    H, W = f0.shape
    report(simulate(f0.reshape(-1), f1.reshape(-1), W, H))


if __name__ == "__main__":
    main()
