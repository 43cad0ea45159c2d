"""Time of the mixed-precision 3x3 weight gradient (conv_wgrad16_kernel + its reduce pass) on the network's layer shapes,
with the wide (64 ci x 128 co) workgroup and without (dsg_set_tuning key 29), and the largest difference between the two.
Usage: DSG_TESTING=1 wgrad16_bench.py [bf16|fp16] [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops, _lib
dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lib = _lib.load()
for (cin, cout, h, w) in [(128, 128, 256, 256), (128, 128, 128, 128), (256, 128, 128, 128), (384, 128, 128, 128), (256, 256, 64, 64), (384, 256, 64, 64), (512, 256, 64, 64), (768, 256, 64, 64),
                          (256, 256, 32, 32), (512, 512, 32, 32), (1024, 512, 32, 32)]:
    x = torch.randn(b, cin // 8, h, w, 8, device="cuda").to(dt)
    dy = (torch.randn(b, cout // 8, h, w, 8, device="cuda") * 1e-2).to(dt)
    ss = torch.stack([1 + 0.1 * torch.randn(b, cin, device="cuda"), 0.1 * torch.randn(b, cin, device="cuda")], -1).contiguous()
    res, line = [], f"  {cin}->{cout} @{h}x{w} b{b}:"
    for wide in (0, 1):
        _lib.check(lib.dsg_set_tuning(29, wide))
        dw = torch.zeros(cout, cin, 3, 3, device="cuda")
        sums = torch.zeros(b, cout, device="cuda")
        bg = torch.zeros(cout, device="cuda")
        ops.conv_wgrad(x, dy, dw, ksize=3, gn_scale_shift=ss, silu=True, dy_sums=sums)
        res.append((dw.clone(), sums.clone()))
        for with_sums in (False, True):  # (the training tape always asks for the dY sums and the bias gradient)
            kw = dict(dy_sums=sums, bias_grad=bg) if with_sums else {}
            for _ in range(3): ops.conv_wgrad(x, dy, dw, ksize=3, gn_scale_shift=ss, silu=True, **kw)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): ops.conv_wgrad(x, dy, dw, ksize=3, gn_scale_shift=ss, silu=True, **kw)
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
            line += f"  {'wide' if wide else '64x64'}{'+sums' if with_sums else ''} {t*1e6:.0f} us"
    d = float((res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max())
    ds = float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max())
    print(line + f"  | rel max diff dw {d:.2e} dy sums {ds:.2e}", flush=True)

# the pointwise gradients (shortcuts: raw sources): the kernel of their own (key 30) against the 3x3 kernel's one-tap instantiation
for (cin, cout, h, w) in [(128, 64, 256, 256), (192, 64, 256, 256), (256, 128, 128, 128), (384, 128, 128, 128), (512, 256, 64, 64),
                          (768, 256, 64, 64), (256, 256, 32, 32)]:
    x = torch.randn(b, cin // 8, h, w, 8, device="cuda").to(dt)
    dy = (torch.randn(b, cout // 8, h, w, 8, device="cuda") * 1e-2).to(dt)
    res, line = [], f"  1x1 {cin}->{cout} @{h}x{w} b{b}:"
    for own in (0, 1):
        _lib.check(lib.dsg_set_tuning(30, own))
        dw = torch.zeros(cout, cin, 1, 1, device="cuda")
        sums = torch.zeros(b, cout, device="cuda")
        bg = torch.zeros(cout, device="cuda")
        ops.conv_wgrad(x, dy, dw, ksize=1, dy_sums=sums)
        res.append((dw.clone(), sums.clone()))
        for _ in range(3): ops.conv_wgrad(x, dy, dw, ksize=1, dy_sums=sums, bias_grad=bg)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ops.conv_wgrad(x, dy, dw, ksize=1, dy_sums=sums, bias_grad=bg)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
        line += f"  {'own tiles' if own else 'one-tap 64x64'} {t*1e6:.0f} us ({2*(cin+cout)*b*h*w/t/1e12:.2f} TB/s of tensors)"
    _lib.check(lib.dsg_set_tuning(30, 1))
    d = float((res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max())
    ds = float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max())
    print(line + f"  | rel max diff dw {d:.2e} dy sums {ds:.2e}", flush=True)
