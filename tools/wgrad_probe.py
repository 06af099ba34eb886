"""Times bflow_conv_wgrad_halo and bflow_grad_stats on the shapes of the DSEC training step (batch 3, crop 288x384) -- tool, not product."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from bflow_amd import split as S

def ev(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = torch.device("cuda:0")
shapes = [  # (B, cin, cout, H, W, kh, kw)   fnet: 15 images; cnet: 3
    (15, 64, 64, 144, 192, 3, 3), (15, 64, 96, 72, 96, 3, 3), (15, 96, 96, 72, 96, 3, 3), (15, 96, 128, 36, 48, 3, 3), (15, 128, 128, 36, 48, 3, 3),
    (3, 64, 64, 144, 192, 3, 3), (3, 128, 128, 36, 48, 3, 3), (3, 256, 192, 36, 48, 3, 3), (3, 128, 256, 36, 48, 3, 3),
    (3, 384, 128, 36, 48, 1, 5), (3, 384, 128, 36, 48, 5, 1), (3, 128, 256, 36, 48, 1, 1), (15, 128, 256, 36, 48, 1, 1), (3, 576, 256, 36, 48, 1, 1)]
for (B, cin, cout, H, W, kh, kw) in shapes:
    x = torch.randn(B, cin, H, W, device=dev); g = torch.randn(B, cout, H, W, device=dev)
    xs, gs = S.from_nchw(x), S.from_nchw(g)
    t = ev(lambda: S.conv_wgrad_halo(xs, gs, cout, cin, (kh, kw)))
    tz = 0.0
    tg = ev(lambda: S.grad_stats(g, 8192.0))
    tp = ev(lambda: S.pow2_scale(g, 8192.0))
    fl = 2.0 * B * H * W * cin * cout * kh * kw
    print(f"B={B:2d} {cin:3d}->{cout:3d} {H}x{W} {kh}x{kw}: wgrad {t:7.1f} us (+ finish)  {fl / t / 1e6:6.1f} TFLOP/s fp32-equiv   grad_stats {tg:6.1f} us pow2 {tp:6.1f} us ({g.numel() * 4 / 1e6:.1f} MB)")
