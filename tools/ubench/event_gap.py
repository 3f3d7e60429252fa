#!/usr/bin/env python3
"""What an event record / a cross-stream wait between two dependent kernels costs on the stream (gap between the
kernels, from a rocprofv3 kernel trace is the better tool; this prints HIP-event spans of 200 back-to-back pairs)."""
import torch, sys
dev = torch.device("cuda", 0)
x = torch.zeros(64 << 20, device=dev)          # 256 MB: ~60 us per pass
y = torch.zeros(1 << 20, device=dev)
S = torch.cuda.Stream(dev)
M = torch.cuda.current_stream(dev)

def run(mode, reps=200):
    torch.cuda.synchronize()
    big = torch.zeros(4096, 4096, device=dev)
    for _ in range(3):
        big = big @ big                              # the host gets ahead
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        y.add_(1.0)
        if mode == "record":
            e = torch.cuda.Event(); e.record(M)
        elif mode == "record+wait_other":
            e = torch.cuda.Event(); e.record(M)
            S.wait_event(e)
            with torch.cuda.stream(S):
                y2.add_(1.0)
        elif mode == "wait_done_event":
            M.wait_event(done)
        y.add_(1.0)
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / reps

y2 = torch.zeros(1 << 20, device=dev)
done = torch.cuda.Event()
with torch.cuda.stream(S):
    y2.add_(1.0)
    done.record(S)
torch.cuda.synchronize()
for mode in ["none", "record", "record+wait_other", "wait_done_event", "none"]:
    print("%-20s %.2f us per pair of 4 MB add_ kernels" % (mode, run(mode)))
