"""What plain streaming kernels reach on this GPU (torch's own fill / copy / reduction over 4 GiB), to read
the roofline fractions of the write-bound and read-bound kernels against: hbm_ceiling.py"""
import torch


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    dev = torch.device("cuda", 0)
    n = 1 << 30  # int32 elements: 4 GiB
    x = torch.empty(n, dtype=torch.int32, device=dev)
    y = torch.empty(n, dtype=torch.int32, device=dev)
    nbytes = 4.0 * n
    t = timed(lambda: x.fill_(1))
    print(f"fill   (write only) {nbytes / t / 1e9:8.0f} GB/s")
    t = timed(lambda: y.copy_(x))
    print(f"copy   (read+write) {2 * nbytes / t / 1e9:8.0f} GB/s")
    t = timed(lambda: x.sum())
    print(f"sum    (read only)  {nbytes / t / 1e9:8.0f} GB/s")
    xb = x.view(torch.uint8)
    t = timed(lambda: torch.count_nonzero(xb))
    print(f"count_nonzero bytes (read only) {nbytes / t / 1e9:8.0f} GB/s")


if __name__ == "__main__":
    main()
