import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python")); sys.path.insert(0, str(ROOT / "oracle"))
import torch, _oflk
import oflk_oracle as O
import oflk_tolerant_model as M
from oflk_synth import synth_pair
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
for (H, W) in ((241, 323), (240, 322)):
    p, c = synth_pair(H, W, 3)
    pp = O.build_gaussian_pyramid(p, 2); pc = O.build_gaussian_pyramid(c, 2)
    for arith in (2, 0):
        plan = _oflk.Plan(0, 1, H, W, 2, 5, 1); plan.set_arithmetic(arith)
        tp, tc = torch.from_numpy(p[None].copy()).to(dev), torch.from_numpy(c[None].copy()).to(dev)
        u = torch.empty_like(tp); v = torch.empty_like(tp)
        plan.pyramidal(tp.data_ptr(), tc.data_ptr(), u.data_ptr(), v.data_ptr(), st); torch.cuda.synchronize()
        u0, v0 = plan.read_level_flow(0, 0, pp[0].shape, st)
        spec = M.tolerant_spec(1, 1) if arith == 2 else M.Spec(1, 1)
        mu0, mv0, _, _ = M.pyramidal(pp[0], pc[0], spec, 5)
        print((H, W), "arith", arith, "level-0 flow bad", int((~((u0 == mu0) & (v0 == mv0))).sum()))
        # upsample of the plan's own level-0 flow, then level 1 by the model
        uu, vv = O.upsample_flow(u0, v0, (H, W))
        spec2 = M.tolerant_spec(2, 1) if arith == 2 else M.Spec(2, 1)
        mu, mv, _, _ = M.pyramidal(p, c, spec2, 5)
        bad = ~((u.cpu().numpy()[0] == mu) & (v.cpu().numpy()[0] == mv))
        print("   final bad", int(bad.sum()))
