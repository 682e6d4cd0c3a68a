// Post-capture surgery on a captured hipGraph (training step, harness/train.py): every MEMSET node becomes a fill KERNEL node
// with the same predecessors and successors.
//
// Why: a captured cfg-4 training step is one chain of ~4 800 nodes, 34 of them memset nodes that the libraries in the step issue
// (ATen's multi-block reductions zero their block semaphore with hipMemsetAsync, aten/src/ATen/native/cuda/Reduce.cuh; MIOpen's
// composable-kernel weight- / data-gradient solvers zero their split-K output).  On this ROCm stack those nodes are not reliably
// ordered with the kernel nodes around them once the graph is that long: from the second replay on the reductions behind them
// returned stale values (the loss's valid-pixel count, the metrics) and the gradients behind the CK memsets came back as zeros —
// the "replays go wrong beside other GPU work" of DESIGN.md §5 (tools/train_graph_ddp_check.py reproduces it with one trainer).
// Kernel nodes are ordered by the queue itself.
#include <vector>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void graph_fill_kernel(unsigned char* dst, unsigned value, unsigned elem, unsigned long long width,
                                                         unsigned long long height, unsigned long long pitch) {
  const unsigned long long total = width * height;
  for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * 256ull) {
    const unsigned long long r = i / width, c = i - r * width;
    unsigned char* p = dst + r * pitch + c * elem;
    if (elem == 4) *reinterpret_cast<unsigned*>(p) = value;
    else if (elem == 2) *reinterpret_cast<unsigned short*>(p) = (unsigned short)value;
    else *p = (unsigned char)value;
  }
}

// one lane writes the device's constant-rate wall clock (100 MHz on gfx950) into its slot: a timeline marker that is an ordinary
// kernel node of a captured graph, ordered like any other launch of its stream
__global__ void stamp_kernel(unsigned long long* buf, int slot) { buf[slot] = wall_clock64(); }

}  // namespace

extern "C" {

/* Timeline marker: buf[slot] = the device wall clock (wall_clock64: constant 100 MHz, 10 ns ticks) when the stream reaches this
 * point.  A kernel launch like any other, so it can be captured into the forward's hipGraph: models/base.py places markers at the
 * phase boundaries of a pass and bench.py reports the phases of a REPLAYED graph without a profiler attached (rocprofv3's queue
 * interception changes how the graph's parallel branches are fed). */
int as_stamp(unsigned long long* buf, int slot, void* stream) {
  AS_REQUIRE(buf && slot >= 0, AS_ERR_BAD_ARG, "stamp: null buffer or negative slot");
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), buf, slot);
  return as::check_launch("stamp");
}

/* graph: the hipGraph_t of a finished stream capture, not yet instantiated (or to be instantiated again afterwards).
 * Replaces each memset node by a fill kernel node (same edges); counts what it replaced and the memset nodes it had to leave
 * (unreadable / unsupported parameters).  Returns AS_OK or an error code. */
int as_graph_replace_memsets(void* graph_, int* replaced, int* left) {
  AS_REQUIRE(graph_, AS_ERR_BAD_ARG, "graph_replace_memsets: null graph");
  hipGraph_t graph = reinterpret_cast<hipGraph_t>(graph_);
  size_t n = 0;
  AS_REQUIRE(hipGraphGetNodes(graph, nullptr, &n) == hipSuccess, AS_ERR_LAUNCH, "graph_replace_memsets: hipGraphGetNodes failed");
  std::vector<hipGraphNode_t> nodes(n);
  if (n) AS_REQUIRE(hipGraphGetNodes(graph, nodes.data(), &n) == hipSuccess, AS_ERR_LAUNCH, "graph_replace_memsets: hipGraphGetNodes failed");
  int done = 0, kept = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType type;
    if (hipGraphNodeGetType(nodes[i], &type) != hipSuccess || type != hipGraphNodeTypeMemset) continue;
    hipMemsetParams mp{};
    if (hipGraphMemsetNodeGetParams(nodes[i], &mp) != hipSuccess || !mp.dst || !(mp.elementSize == 1 || mp.elementSize == 2 || mp.elementSize == 4) ||
        mp.width == 0 || mp.width > (1ull << 40) || mp.height > (1ull << 32)) {
      ++kept;
      continue;
    }
    size_t nd = 0, ns = 0;
    if (hipGraphNodeGetDependencies(nodes[i], nullptr, &nd) != hipSuccess || hipGraphNodeGetDependentNodes(nodes[i], nullptr, &ns) != hipSuccess) { ++kept; continue; }
    std::vector<hipGraphNode_t> deps(nd), succ(ns);
    if (nd && hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd) != hipSuccess) { ++kept; continue; }
    if (ns && hipGraphNodeGetDependentNodes(nodes[i], succ.data(), &ns) != hipSuccess) { ++kept; continue; }
    unsigned char* dst = static_cast<unsigned char*>(mp.dst);
    unsigned value = mp.value, elem = mp.elementSize;
    unsigned long long width = mp.width, height = mp.height ? mp.height : 1, pitch = mp.pitch;
    if (elem == 1 && height == 1 && (width & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0) {  // bytes -> words
      value &= 0xFFu;
      value |= value << 8;
      value |= value << 16;
      elem = 4;
      width >>= 2;
    }
    const unsigned long long total = width * height;
    unsigned long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    void* args[] = {&dst, &value, &elem, &width, &height, &pitch};
    hipKernelNodeParams kp{};
    kp.func = reinterpret_cast<void*>(graph_fill_kernel);
    kp.gridDim = dim3((unsigned)blocks);
    kp.blockDim = dim3(256);
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
    hipGraphNode_t fill;
    const hipError_t e = hipGraphAddKernelNode(&fill, graph, nd ? deps.data() : nullptr, nd, &kp);
    if (e != hipSuccess) return as::fail(AS_ERR_LAUNCH, "graph_replace_memsets: hipGraphAddKernelNode: %s", hipGetErrorString(e));
    for (size_t k = 0; k < ns; ++k) {
      const hipError_t e2 = hipGraphAddDependencies(graph, &fill, &succ[k], 1);
      if (e2 != hipSuccess) return as::fail(AS_ERR_LAUNCH, "graph_replace_memsets: hipGraphAddDependencies: %s", hipGetErrorString(e2));
    }
    const hipError_t e3 = hipGraphDestroyNode(nodes[i]);
    if (e3 != hipSuccess) return as::fail(AS_ERR_LAUNCH, "graph_replace_memsets: hipGraphDestroyNode: %s", hipGetErrorString(e3));
    ++done;
  }
  if (replaced) *replaced = done;
  if (left) *left = kept;
  return AS_OK;
}

}  // extern "C"
