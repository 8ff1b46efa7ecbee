"""What one dependent kernel node of a replayed hipGraph costs on this box: a chain of n trivial launches (a 1 KB fill)
captured once, replayed 200 times; also the same chain with every other node a 20 us kernel (does the floor hide behind
a long neighbour?).  python tools/graph_node_floor.py"""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(256, device=dev)
big = torch.zeros(64 * 1024 * 1024 // 4, device=dev)    # 64 MB fill: ~15-20 us


def measure(fn, reps=200):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


for n in (1, 10, 50, 100):
    t = measure(lambda: [x.fill_(1.0) for _ in range(n)])
    print("chain of %3d trivial nodes: %8.1f us per replay = %.2f us per node" % (n, t, t / n))
tb = measure(lambda: [big.fill_(1.0) for _ in range(20)])
print("20 x 64 MB fills: %.1f us per replay = %.2f us each" % (tb, tb / 20))
tm = measure(lambda: [(big.fill_(1.0), x.fill_(1.0)) for _ in range(20)])
print("20 x (64 MB fill + trivial node): %.1f us per replay -> a trivial node behind a long one costs %.2f us" % (tm, (tm - tb) / 20))
