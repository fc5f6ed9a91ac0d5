"""TEST INFRASTRUCTURE (oracle of the RTL-bit-accurate integer mode) -- never on
the product path: only tests/ may import this.

Closed form (NumPy, data-parallel) of what the reference's single-scale RTL computes and of what its
testbench samples, derived from -- and held equal to, tests/test_rtl_model.py -- the literal cycle-by-cycle
execution in oracle/rtl_cycle_sim.py.  PARITY UNPINNED, for the reason stated there (no simulator here, no
output of the RTL as committed in the reference).

The arithmetic, as the RTL has it (not as the Python reference has it):

* frame_buffer_simple streams one pixel per clock with `valid` raised one clock early: the line buffers
  ingest a stream S of W*H + 1 values, S[0] = 0, S[n] = pixel n-1 (8 bits, reinterpreted as SIGNED by
  line_buffer_5x5's `logic signed` ports, gradient_compute.sv:63-87).
* line_buffer_5x5.sv:75-151: after ingesting stream position (r, c) the 5x5 window is, for c < W-1,
      rows i = 0..3, columns j = 0..3 : stream positions (r-3+i, c-4+j)
      rows i = 0..3, column  j = 4    : stream positions (r-4+i, c)
      row  i = 4                      : stream positions (r, c-4 .. c)
  (the line memories are written one pixel late and read at the already advanced column: the two skews
  cancel except in the last column, and the newest complete row appears twice), and for c = W-1 rows 0..3
  are zero except column 4 = stream positions (r-4+i, W-1).  `window_valid` = (r >= 4 and c >= 4), registered,
  and it HOLDS while `data_valid` is low; window_x / window_y = (col - 2, row - 2) of the advanced counters.
* gradient_compute.sv:89-139 on the window's centre 3x3 (a proper neighbourhood of stream position
  (r-1, c-2)): avg = ((curr + prev) mod 512) >> 1 of the SIGNED 8-bit pixels, Sobel sums >>> 3 into 12 bits,
  It = prev - curr of the unsigned pixels.
* window_accumulator.sv:100-189: the gradients of the valid windows form a second stream (W-4 per image row)
  that enters line buffers of the SAME width W, so its rows are not image rows; 25 products of 12-bit values
  per quantity, 32-bit sums.
* flow_solver.sv:82-149: the six products of sums truncated to their low 32 bits, det and numerators as
  32-bit differences, |det| > 1000, (num <<< 7) / det in 39 bits truncating towards zero, low 16 bits,
  clamp to +-1024 (S8.7).
* tb/tb_optical_flow_top.sv:176-240 samples flow_valid every clock until `done`: a state is sampled once per
  clock it is held (the four-clock gap at every image row's start repeats the last vector four times), and the
  pipeline's last TAIL_CLOCKS clocks fall after `done`.
"""
import numpy as np

TAIL_CLOCKS = 5   # stream positions at the end of the frame whose flow the testbench no longer samples (calibrated against the cycle simulation)


def _s8(a):
    a = np.asarray(a, np.int64) & 0xFF
    return np.where(a >= 128, a - 256, a)


def _sx(a, bits):
    a = np.asarray(a, np.int64) & ((1 << bits) - 1)
    return np.where(a >> (bits - 1), a - (1 << bits), a)


def gradient_stream(f0, f1):
    """(gx, gy, gt, n) of the valid gradient windows in stream order; f0 = previous frame, f1 = current frame,
    uint8 [H, W]; n = their stream positions"""
    H, W = f0.shape
    total = H * W
    s_prev = np.concatenate([[0], f0.reshape(-1).astype(np.int64)])   # stream: one leading bogus value
    s_curr = np.concatenate([[0], f1.reshape(-1).astype(np.int64)])
    r, c = np.divmod(np.arange(total), W)
    n = np.nonzero((r >= 4) & (c >= 4))[0]
    r, c = r[n], c[n]
    edge = c == W - 1

    def tap(s, i, j):     # centre 3x3 of the window, i, j = 0..2: stream position (r-2+i, c-3+j); zero in the last column
        v = s[(r - 2 + i) * W + (c - 3 + j)]
        return np.where(edge, 0, v)

    avg = [[((_s8(tap(s_curr, i, j)) + _s8(tap(s_prev, i, j))) & 0x1FF) >> 1 for j in range(3)] for i in range(3)]
    xl = -avg[0][0] - (avg[1][0] << 1) - avg[2][0]
    xr = avg[0][2] + (avg[1][2] << 1) + avg[2][2]
    gx = _sx((xl + xr) >> 3, 12)
    yt = -avg[0][0] - (avg[0][1] << 1) - avg[0][2]
    yb = avg[2][0] + (avg[2][1] << 1) + avg[2][2]
    gy = _sx((yt + yb) >> 3, 12)
    gt = _sx((tap(s_prev, 1, 1) & 0xFF) - (tap(s_curr, 1, 1) & 0xFF), 12)
    return gx, gy, gt, n


def window_indices(k, W):
    """[len(k), 25] indices into the gradient stream of the 5x5 window after ingesting stream element k (row-major
    i, j); -1 = zero"""
    k = np.asarray(k, np.int64)
    r2, c2 = np.divmod(k, W)
    edge = c2 == W - 1
    idx = np.empty((len(k), 25), np.int64)
    for i in range(5):
        for j in range(5):
            if i == 4:
                v = k - (4 - j)
            elif j == 4:
                v = np.where(edge, (r2 - 3 + i) * W - 1, k - (4 - i) * W)
            else:
                v = np.where(edge, -1, k - (3 - i) * W - (4 - j))
            idx[:, i * 5 + j] = np.where(v < 0, -1, v)
    return idx


def trunc_div(a, b):
    q = np.abs(a) // np.abs(b)
    return np.where((a >= 0) == (b >= 0), q, -q)


def flow_states(gx, gy, gt, W):
    """per element k of the gradient stream: (valid, x, y, u, v) after the accumulator ingested it"""
    M = len(gx)
    k = np.arange(M, dtype=np.int64)
    r2, c2 = np.divmod(k, W)
    valid = (r2 >= 4) & (c2 >= 4)
    col, row = np.where(c2 == W - 1, 0, c2 + 1), np.where(c2 == W - 1, r2 + 1, r2)
    has_xy = valid & (col >= 2) & (row >= 2)
    x, y = np.where(has_xy, col - 2, 0), np.where(has_xy, row - 2, 0)
    idx = window_indices(k, W)
    pad = lambda g: np.concatenate([g.astype(np.int64), [0]])   # index -1 -> 0
    wx, wy, wt = pad(gx)[idx], pad(gy)[idx], pad(gt)[idx]
    sxx, syy, sxy, sxt, syt = ((a * b).sum(axis=1) for a, b in ((wx, wx), (wy, wy), (wx, wy), (wx, wt), (wy, wt)))
    lo = lambda p: _sx(p, 32)
    det = _sx(lo(sxx * syy) - lo(sxy * sxy), 32)
    nu = _sx(lo(syy * sxt) - lo(sxy * syt), 32)
    nv = _sx(lo(sxx * syt) - lo(sxy * sxt), 32)
    solvable = (det > 1000) | (det < -1000)
    d = np.where(solvable, det, 1)
    u = np.clip(_sx(trunc_div(nu << 7, d), 16), -1024, 1024)
    v = np.clip(_sx(trunc_div(nv << 7, d), 16), -1024, 1024)
    return valid, x, y, np.where(solvable, u, 0), np.where(solvable, v, 0)


def testbench_vectors(f0, f1):
    """[N, 4] int64 (flow_x, flow_y, flow_u, flow_v) in the order the testbench's monitor loop samples them"""
    H, W = f0.shape
    gx, gy, gt, n = gradient_stream(f0, f1)
    valid, x, y, u, v = flow_states(gx, gy, gt, W)
    # one sample per clock of the frame: the state of the last gradient ingested at or before stream position p
    p = np.arange(H * W - TAIL_CLOCKS)
    k = np.searchsorted(n, p, side="right") - 1
    k = k[k >= 0]
    k = k[valid[k]]
    return np.stack([x[k], y[k], u[k], v[k]], axis=1)
