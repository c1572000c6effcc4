"""What this MI355X sustains for plain streaming kernels (torch elementwise): fill (write only), copy (read + write), sum (read only), over
buffers far larger than the 256 MB memory-side cache."""
import torch
dev = "cuda"
N = 1 << 28      # 1 GiB of fp32
a, b = torch.empty(N, device=dev), torch.empty(N, device=dev)


def t(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gb = N * 4 / 1e12
print("fill  (write)      %.2f TB/s" % (gb / t(lambda: a.fill_(1.0))))
print("copy  (read+write) %.2f TB/s of traffic" % (2 * gb / t(lambda: b.copy_(a))))
print("sum   (read)       %.2f TB/s" % (gb / t(lambda: a.sum())))
print("mul_  (read+write in place) %.2f TB/s of traffic" % (2 * gb / t(lambda: a.mul_(1.0001))))
