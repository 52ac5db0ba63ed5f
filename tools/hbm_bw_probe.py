"""What the box's HBM does for plain streams (torch fill / copy / sum of 8 GiB): the yardstick for the stage kernels' byte rates (profiles/r03w_work_order.txt (8)).

  gpurun -- python tools/hbm_bw_probe.py
"""
import torch, time
x = torch.empty(8 << 30, dtype=torch.uint8, device='cuda')
y = torch.empty(8 << 30, dtype=torch.uint8, device='cuda')
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: x.zero_()); print("memset 8 GiB: %.3f ms = %.2f TB/s written" % (ms, 8.59 / ms))
ms = t(lambda: y.copy_(x)); print("copy 8 GiB: %.3f ms = %.2f TB/s read + %.2f TB/s written" % (ms, 8.59 / ms, 8.59 / ms))
xf = x.view(torch.float32)
ms = t(lambda: xf.sum()); print("sum 8 GiB: %.3f ms = %.2f TB/s read" % (ms, 8.59 / ms))
