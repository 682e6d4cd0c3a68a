"""Node inventory of the captured cfg-4 training step (run on the GPU box): captures the whole-step hipGraph with the hipGraph_t
kept, walks its nodes through the HIP runtime (hipGraphGetNodes / NodeGetType / NodeGetDependencies) and prints the histogram of
node kinds plus, for every memcpy node, the kernels in front of and behind it (which operator issued the copy).

    python tools/train_graph_nodes.py
"""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))

KINDS = ["Kernel", "Memcpy", "Memset", "Host", "Graph", "Empty", "WaitEvent", "EventRecord", "ExtSemSignal", "ExtSemWait", "MemAlloc",
         "MemFree", "MemcpyFromSymbol", "MemcpyToSymbol", "BatchMemOp"]
CPY = {0: "H2H", 1: "H2D", 2: "D2H", 3: "D2D", 4: "Default"}


class Dim3(C.Structure):
    _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("z", C.c_uint)]


class KernelParams(C.Structure):
    _fields_ = [("blockDim", Dim3), ("extra", C.c_void_p), ("func", C.c_void_p), ("gridDim", Dim3), ("kernelParams", C.c_void_p),
                ("sharedMemBytes", C.c_uint)]


def main():
    import torch
    from anystereo.harness.synthetic import fill_module_deterministic
    from anystereo.harness.train import Trainer, synthetic_train_batch
    from anystereo.models import __models__, default_args
    os.environ["ANYSTEREO_TRAIN_GRAPH_KEEP"] = "1"
    args = default_args("continuous_IGEVStereo")
    m = __models__["continuous_IGEVStereo"](args)
    fill_module_deterministic(m, base_seed=1)
    tr = Trainer(m.to("cuda:0"), train_iters=16, max_disp=args.max_disp, graph=True)
    batch = synthetic_train_batch(4, 160, 320, seed=0, device="cuda:0")
    for _ in range(tr.graph_warmup + 2):
        loss, _ = tr.step(batch)
    torch.cuda.synchronize()
    print("loss after two replays:", float(loss))
    graph = C.c_void_p(int(tr._graph["graph"].raw_cuda_graph()))
    hip = C.CDLL("libamdhip64.so")
    hip.hipKernelNameRefByPtr.restype = C.c_char_p
    hip.hipKernelNameRefByPtr.argtypes = [C.c_void_p, C.c_void_p]
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(graph, None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(graph, nodes, C.byref(n)) == 0
    hist = collections.Counter()
    copies = []
    for nd in nodes:
        t = C.c_int(-1)
        hip.hipGraphNodeGetType(C.c_void_p(nd), C.byref(t))
        kind = KINDS[t.value] if 0 <= t.value < len(KINDS) else f"type{t.value}"
        hist[kind] += 1
        if kind == "Memcpy":
            copies.append(nd)
    print(f"{n.value} nodes:", dict(hist))

    def kname(nd):
        kp = KernelParams()
        if hip.hipGraphKernelNodeGetParams(C.c_void_p(nd), C.byref(kp)) != 0:
            return "?"
        s = hip.hipKernelNameRefByPtr(C.c_void_p(kp.func), None)
        return (s or b"?").decode(errors="replace")[:110]
    # shape of the dependency graph: a single-stream capture is ONE chain (every node one predecessor, one successor)
    def count(nd, fn):
        k = C.c_size_t(0)
        return k.value if fn(C.c_void_p(nd), None, C.byref(k)) == 0 else -1
    roots, leaves, forks, joins = [], [], [], []
    for nd in nodes:
        nin, nout = count(nd, hip.hipGraphNodeGetDependencies), count(nd, hip.hipGraphNodeGetDependentNodes)
        if nin == 0:
            roots.append(nd)
        if nout == 0:
            leaves.append(nd)
        if nout > 1:
            forks.append(nd)
        if nin > 1:
            joins.append(nd)

    def label(nd):
        t = C.c_int(-1)
        hip.hipGraphNodeGetType(C.c_void_p(nd), C.byref(t))
        return kname(nd)[:90] if t.value == 0 else KINDS[t.value]
    print(f"dependency shape: {len(roots)} root(s), {len(leaves)} leaf node(s), {len(forks)} node(s) with more than one successor, {len(joins)} with more than one predecessor")
    for title, lst in (("roots", roots), ("leaves", leaves), ("forks", forks), ("joins", joins)):
        for nd in lst[:12]:
            print(f"  {title[:-1]}: {label(nd)}")
    summary = collections.Counter()

    def neighbours(nd, fn):
        k = C.c_size_t(0)
        out = []
        if fn(C.c_void_p(nd), None, C.byref(k)) == 0 and k.value:
            arr = (C.c_void_p * k.value)()
            fn(C.c_void_p(nd), arr, C.byref(k))
            for d in arr:
                t = C.c_int(-1)
                hip.hipGraphNodeGetType(C.c_void_p(d), C.byref(t))
                out.append(kname(d)[:160] if t.value == 0 else KINDS[t.value])
        return out
    msum = collections.Counter()
    for nd in nodes:
        t = C.c_int(-1)
        hip.hipGraphNodeGetType(C.c_void_p(nd), C.byref(t))
        if t.value == 2:
            msum[(" | ".join(neighbours(nd, hip.hipGraphNodeGetDependencies)), " | ".join(neighbours(nd, hip.hipGraphNodeGetDependentNodes)))] += 1
    print("memset nodes by (producer kernel) -> (consumer kernel):")
    for (pre, post), v in msum.most_common():
        print(f"  m{v:3d}  after [{pre[:120]}]  before [{post[:160]}]")
    for nd in copies:
        summary[(" | ".join(neighbours(nd, hip.hipGraphNodeGetDependencies)), " | ".join(neighbours(nd, hip.hipGraphNodeGetDependentNodes)))] += 1
    print("memcpy nodes by (producer kernels) -> (consumer kernels)  [hipGraphMemcpyNodeGetParams returns nothing usable for captured 1-D copies]:")
    for (pre, post), v in summary.most_common():
        print(f"  x{v:3d}  after [{pre[:150]}]  before [{post[:150]}]")


if __name__ == "__main__":
    main()
