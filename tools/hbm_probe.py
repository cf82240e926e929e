"""What the box actually delivers (peaks to be confirmed on the GPU): CU count and clocks from rocminfo,
attainable HBM bandwidth from device-to-device copies and a read-only reduction of a 4 GiB buffer."""
import subprocess
import time

import torch

dev = torch.device("cuda:0")
info = subprocess.run(["rocminfo"], capture_output=True, text=True).stdout
for key in ("Marketing Name", "Compute Unit", "Max Clock Freq", "Name:                    gfx"):
    for line in info.splitlines():
        if key in line and ("gfx" in info[info.find(line) - 2000:info.find(line) + 2000]):
            print(line.strip())
            break
n = 1 << 30                      # 4 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device=dev).fill_(1.0)
b = torch.empty_like(a)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timed(lambda: b.copy_(a))
print("d2d copy 4 GiB: %.2f ms -> %.2f TB/s (read + write)" % (t * 1e3, 2 * 4 * n / t / 1e12))
t = timed(lambda: a.sum())
print("read-only reduction 4 GiB: %.2f ms -> %.2f TB/s" % (t * 1e3, 4 * n / t / 1e12))
t = timed(lambda: b.fill_(2.0))
print("write-only fill 4 GiB: %.2f ms -> %.2f TB/s" % (t * 1e3, 4 * n / t / 1e12))
