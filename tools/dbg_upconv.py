import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
from test_gpu_kernels import *
prec = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
B, H, W, Ci, Co = 2, 8, 8, 128, 64
g = torch.Generator().manual_seed(4)
x = _round(torch.randn(B, Ci, H, W, generator=g), prec)
w = torch.randn(Ci, Co, 2, 2, generator=g) / Ci ** 0.5
b = torch.randn(Co, generator=g)
i16 = dict(dtype=torch.int16, device="cuda")
n = 4 * Ci * Co
fh, fl, dh, dl = (torch.empty(n, **i16) for _ in range(4))
wd = w.cuda(); bd = b.cuda()
call("crimac_pack_upconv2x2", ptr(wd), Ci, Co, ptr(fh), ptr(fl), ptr(dh), ptr(dl))
torch.cuda.synchronize()
ref = F.conv_transpose2d(x, _round(w, prec), b, stride=2)
cat = torch.zeros(B * 4 * H * W, 2 * Co, dtype=_dt(prec), device="cuda")
xn = to_nhwc(x, prec)
torch.cuda.synchronize()
call("crimac_igemm_conv", hip.PREC_NAMES[prec], ptr(xn), Ci, B, H, W, H, W, Ci, 4 * Co, 1, 1, 0, 1,
     ptr(fh), ptr(fl), ptr(bd), Co, ptr(cat), 2 * Co, 0, 1, Co)
torch.cuda.synchronize()
out = from_nhwc(cat[:, :Co].contiguous(), B, 2 * H, 2 * W)
bad = (out - ref).abs() > 1e-3
print("bad count", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()
print(idx[:40].tolist())
# per (b, y, x) pixel count
print("bad per batch", bad.sum((1,2,3)).tolist())
print("bad channels", bad.sum((0,2,3)).tolist())
print("bad rows y", bad.sum((0,1,3)).tolist())
print("bad cols x", bad.sum((0,1,2)).tolist())
