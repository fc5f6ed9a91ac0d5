#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; GPU sanitizers are not available on
# the pool): edge shapes (1x1 ... 240x320), windows 1 ... 27, 1-4 levels, 0-3 iterations through every oracle entry point.
# Usage: bash tools/oracle_sanitize.sh   (run in the build container; needs gcc's libasan / libubsan)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cp -r "$R/oracle" "$T/"
gcc -O1 -g -fPIC -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o "$T/oracle/liboflk_oracle.so" "$T/oracle/oflk_oracle.c" -lm
cat > "$T/run.py" <<PY
import sys, numpy as np
sys.path.insert(0, "$T/oracle")
sys.path.insert(0, "$R/optical-flow-fpga_amd/python")
import oflk_oracle as O
rng = np.random.default_rng(5)
n = 0
for (H, W) in [(1, 1), (2, 3), (5, 7), (17, 33), (48, 64), (97, 131), (240, 320)]:
    for win in (1, 3, 4, 5, 7, 11, 13, 27):
        a = rng.uniform(0, 255, (H, W)).astype(np.float32)
        b = np.roll(a, 1, 1) + rng.normal(0, 1, (H, W)).astype(np.float32)
        O.lucas_kanade_single_scale(a, b, win)
        gx, gy, gt = O.compute_gradients(a, b)
        O.lucas_kanade_from_gradients(gx, gy, gt, win)
        for L in (1, 2, 3, 4):
            if int(H * 0.5 ** (L - 1)) < 1 or int(W * 0.5 ** (L - 1)) < 1:
                continue
            for K in (0, 1, 3):
                O.lucas_kanade_pyramidal_ex(a, b, L, win, K)
                n += 1
    O.build_gaussian_pyramid(a, 2)
    O.warp_image(a, b * 0.01, a * -0.01)
    O.upsample_flow(a, b, (2 * H + 1, 2 * W - 1 if W > 1 else 2))
print("oracle under ASan + UBSan:", n, "pyramidal cases and every other entry point, no report")
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python3 "$T/run.py"
rm -rf "$T"
