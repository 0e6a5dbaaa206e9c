import sys; sys.path.insert(0, '/root/repo')
import torch, numpy as np
from oracle import glsdet_oracle as O, mpdet_oracle as M
from tests.helpers import calibrated_resdet_sd
from glsdet_amd.resdet import HipGflDetector
x = O.synth_input((1, 3, 128, 160), 7)
pl = HipGflDetector.DEFAULTS["proxies_list"]
for gl in (False, True):
    sd = calibrated_resdet_sd("mpdet", 1, x, gl_fusion=gl) if gl else calibrated_resdet_sd("mpdet", 1, x)
    wc, wr = M.mpdet_forward(sd, x, pl)
    for mode in ("f32", "f16"):
        gc, gr = HipGflDetector("mpdet", sd, dtype=mode).forward_raw(x.cuda())
        print("gl" if gl else "plain", mode, "cls err/max:", ["%.3f/%.2f" % (float((g.cpu() - w).abs().max()), float(w.abs().max())) for g, w in zip(gc, wc)],
              "reg:", ["%.3f/%.2f" % (float((g.cpu() - w).abs().max()), float(w.abs().max())) for g, w in zip(gr, wr)])
    st = M.gl_fusion_inputs(sd, "neck", M.resnet(sd, "backbone", x))
    print("   stage max |feat|:", [round(float(s.abs().max()), 1) for s in st], "rms:", [round(float(s.pow(2).mean().sqrt()), 2) for s in st])
