"""How does a replayed hipGraph schedule two INDEPENDENT chains forked from one node?  (round 6: the pre-loop's two branches —
stems + context network | feature trunk + cost aggregation — run one after the other in the replayed forward,
profiles/r06_base_pass_timeline.json.)  Chains of ~20 us spin kernels (one block each: they cannot contend for CUs), captured in
different ISSUE orders; replay time by HIP events.  parallel = max of the chains, serial = their sum."""
import sys
import torch

dev = torch.device("cuda", 0)
CY = 40000  # ~20 us


def chain(n):
    for _ in range(n):
        torch.cuda._sleep(CY)


def capture(build):
    side = torch.cuda.Stream(device=dev)
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        build(s, side)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        build(torch.cuda.current_stream(), side)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)


NA, NB = 50, 50


def serial(main, side):
    chain(NA + NB)


def fork_side_first(main, side):
    chain(1)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        chain(NA)
    chain(NB)
    main.wait_stream(side)
    chain(1)


def fork_main_first(main, side):
    chain(1)
    ev = torch.cuda.Event()
    ev.record(main)
    chain(NB)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        chain(NA)
    main.wait_stream(side)
    chain(1)


def make_interleaved(group):
    def f(main, side):
        chain(1)
        side.wait_stream(main)
        a = b = 0
        while a < NA or b < NB:
            with torch.cuda.stream(side):
                k = min(group, NA - a)
                chain(k)
                a += k
            k = min(group, NB - b)
            chain(k)
            b += k
        main.wait_stream(side)
        chain(1)
    return f


def fork_join_short_then_long(main, side):
    # side: long; main: short, then join, then long: does main's short part overlap the side's long one?
    chain(1)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        chain(NA)
    chain(5)
    main.wait_stream(side)
    chain(NB)


def three_way(main, side):
    s3 = torch.cuda.Stream(device=dev)
    chain(1)
    side.wait_stream(main)
    s3.wait_stream(main)
    with torch.cuda.stream(side):
        chain(NA)
    with torch.cuda.stream(s3):
        chain(NA)
    chain(NB)
    main.wait_stream(side)
    main.wait_stream(s3)
    chain(1)


cases = [("serial (one stream, %d kernels)" % (NA + NB), serial), ("fork, side chain issued first", fork_side_first),
         ("fork, main chain issued first (event fork)", fork_main_first), ("fork, issue interleaved 1:1", make_interleaved(1)),
         ("fork, issue interleaved 5:5", make_interleaved(5)), ("fork, issue interleaved 25:25", make_interleaved(25)),
         ("side long | main short -> join -> main long", fork_join_short_then_long), ("three chains of 50", three_way)]
unit = None
for name, fn in cases:
    us = capture(fn)
    if unit is None:
        unit = us / (NA + NB)
    print("%-52s %8.1f us  = %.1f kernel units" % (name, us, us / unit))
sys.stdout.flush()
