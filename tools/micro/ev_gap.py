import torch, time
dev = torch.device("cuda:0")
x = torch.ones(192 * 1024 * 1024, device=dev)  # 768 MB: one op reads + writes 1.5 GB (~0.3 ms)
y = torch.ones(1024, device=dev)
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, n=200):
    fn(10); torch.cuda.synchronize()
    t = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6
def same_stream(n):
    with torch.cuda.stream(s1):
        for i in range(n): x.mul_(1.0)
def alternating(n):  # sweep i on stream i % 2, each waits for the previous one's event (today's chain)
    ev = None
    for i in range(n):
        s = (s1, s2)[i % 2]
        if ev is not None: s.wait_event(ev)
        with torch.cuda.stream(s):
            x.mul_(1.0)
            ev = torch.cuda.Event(); ev.record(s)
            y.add_(1.0); y.add_(1.0); y.add_(1.0)   # a tail behind the sweep on its own stream
def sweep_stream(n):  # all sweeps on s3; pre and tail on the callers' streams, events both ways
    for i in range(n):
        s = (s1, s2)[i % 2]
        with torch.cuda.stream(s):
            y.add_(1.0)                       # qprep
            e1 = torch.cuda.Event(); e1.record(s)
        s3.wait_event(e1)
        with torch.cuda.stream(s3):
            x.mul_(1.0)
            e2 = torch.cuda.Event(); e2.record(s3)
        s.wait_event(e2)
        with torch.cuda.stream(s):
            y.add_(1.0); y.add_(1.0); y.add_(1.0)
base = timed(same_stream)
print("same stream back to back: %.1f us per op" % base)
print("alternating streams, event chain: %.1f us per op (+%.1f)" % (timed(alternating), timed(alternating) - base))
print("dedicated sweep stream: %.1f us per op (+%.1f)" % (timed(sweep_stream), timed(sweep_stream) - base))
