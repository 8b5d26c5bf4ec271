"""What HBM streaming rates this box delivers to plain kernels (torch): read-only reductions and
copies over buffers of the sizes the emission / statistics kernels move per launch (94 MB of
frames in, 216 MB of densities out; 310 MB in for the statistics), cold (a 1 GiB scrub between
runs evicts the 256 MB Infinity Cache) and warm.  usage: python profiles/tools/hbm_rate.py"""
import time
import torch

dev = "cuda:0"
scrub = torch.empty(1 << 27, dtype=torch.float64, device=dev)  # 1 GiB


def timed(fn, cold, reps=5):
    ts = []
    for _ in range(reps):
        if cold:
            scrub.fill_(1.0)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return min(ts)


for mb in (94, 216, 310, 1024):
    n = mb * 1000 * 1000 // 8
    x = torch.randn(n, dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    for cold in (True, False):
        t_sum = timed(lambda: x.sum(), cold)
        t_cpy = timed(lambda: y.copy_(x), cold)
        t_fill = timed(lambda: y.fill_(2.0), cold)
        print(f"{mb:5d} MB {'cold' if cold else 'warm'}: read {mb / 1e6 / t_sum:6.2f} TB/s ({t_sum * 1e6:7.1f} us)   "
              f"copy (r+w) {2 * mb / 1e6 / t_cpy:6.2f} TB/s ({t_cpy * 1e6:7.1f} us)   write {mb / 1e6 / t_fill:6.2f} TB/s ({t_fill * 1e6:7.1f} us)")
    del x, y
