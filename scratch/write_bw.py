import torch
n = 1342177280  # 5.4 GB of fp32
x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.fill_(1.0)); print("fill   %.3f ms  %.2f TB/s written" % (ms, n * 4 / ms / 1e9))
ms = t(lambda: x.zero_()); print("zero   %.3f ms  %.2f TB/s written" % (ms, n * 4 / ms / 1e9))
ms = t(lambda: y.copy_(x)); print("copy   %.3f ms  %.2f TB/s read + %.2f TB/s written" % (ms, n * 4 / ms / 1e9, n * 4 / ms / 1e9))
ms = t(lambda: x.sum()); print("sum    %.3f ms  %.2f TB/s read" % (ms, n * 4 / ms / 1e9))
