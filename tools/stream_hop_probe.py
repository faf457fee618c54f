"""What a hop to a second stream and back costs on this part (EXPERIMENTS R5.3): per iteration a tiny kernel, then either a copy
(+ a device-side sleep) in line, or the same on a second stream bracketed by wait_stream / event / wait_event.
    python tools/stream_hop_probe.py"""
import time, torch
x = torch.zeros(1 << 20, device="cuda"); y = torch.zeros_like(x)
side = torch.cuda.Stream()
def run(n, sleep_cycles, use_side):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        x.add_(1.0)
        if use_side:
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                y.copy_(x)
                if sleep_cycles: torch.cuda._sleep(sleep_cycles)
                ev = torch.cuda.Event(); ev.record()
            cur.wait_event(ev)
        else:
            y.copy_(x)
            if sleep_cycles: torch.cuda._sleep(sleep_cycles)
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    return host / n * 1e6, wall / n * 1e6
for n in (20, 100, 400):
    for sl in (0, 10000, 100000):
        for side_ in (False, True):
            h, w = run(n, sl, side_)
            print(f"n {n} sleep {sl} side {side_}: host {h:.1f} us/iter wall {w:.1f}")
