import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import torch, bench
from glsdet_amd.detector import HipDetector
dev="cuda:0"
kind, tag, H, W, bs = bench.WORKLOADS["yolox_s_glfusion_1344x800_bs8"]
img = torch.randn(bs, 3, H, W, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
out = HipDetector(kind, bench.synthetic_state_dict(tag), dtype="f16", device=dev).forward_raw(img)
h = hashlib.sha256()
for o in out: h.update(o.cpu().numpy().tobytes())
print("logits sha256", h.hexdigest()[:16], float(out[0].abs().max()))
