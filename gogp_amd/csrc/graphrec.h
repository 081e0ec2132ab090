// graphrec.h -- an explicitly built hipGraph instead of stream capture.
//
// The sweep of one evaluation forks and joins over six streams; hipStreamEndCapture of this ROCm runtime does not
// survive that pattern (round 2), so round 2-3 could only replay LINEAR graphs (N <= 1024).  Here no stream is ever
// captured: while a recorder is active on the calling thread (tl_rec), every launch of the library's own kernels,
// every async copy / fill and every event record / wait of the orchestration goes to the recorder, which adds a
// node with hipGraphAddKernelNode / hipGraphAddMemcpyNode1D / hipGraphAddMemsetNode and EXPLICIT dependencies:
//   * a stream is only a key: the node last added "on" it is the next node's first dependency (in-order stream);
//   * an event recorded on a stream stands for that stream's current frontier; a stream that waits for it adds
//     that frontier to the dependencies of its next node.
// The orchestration code (api.hip) is the same for the stream path and the graph path; results are bit-identical
// (same kernels, same arguments, a dependency set that contains the stream path's).
#pragma once
#include <hip/hip_runtime.h>

#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

namespace gogp {

struct GraphRec {
  hipGraph_t graph = nullptr;
  std::unordered_map<hipStream_t, hipGraphNode_t> last;                // last node of a stream (absent: none yet)
  std::unordered_map<hipStream_t, std::vector<hipGraphNode_t>> pend;   // frontiers it has waited for since
  std::unordered_map<hipEvent_t, std::vector<hipGraphNode_t>> evnodes; // what an event stands for
  hipError_t err = hipSuccess;
  int nodes = 0;
  // diagnostic form (option graph = 3): ONE chain in enqueue order -- every node depends on the node added before it,
  // whatever its stream (a linear extension of the DAG: the stream path's enqueue order)
  bool chain = false;
  hipGraphNode_t glast = nullptr;
  // dependencies of the next node on stream s; clears the pending list
  std::vector<hipGraphNode_t> deps(hipStream_t s) {
    std::vector<hipGraphNode_t> d;
    if (chain) {
      if (glast) d.push_back(glast);
      return d;
    }
    auto l = last.find(s);
    if (l != last.end() && l->second) d.push_back(l->second);
    auto p = pend.find(s);
    if (p != pend.end()) {
      for (hipGraphNode_t n : p->second) {
        bool dup = false;
        for (hipGraphNode_t m : d) dup = dup || m == n;
        if (!dup) d.push_back(n);
      }
      p->second.clear();
    }
    return d;
  }
  void added(hipStream_t s, hipGraphNode_t n, hipError_t e) {
    if (e != hipSuccess && err == hipSuccess) err = e;
    if (e == hipSuccess) {
      last[s] = n;
      glast = n;
      ++nodes;
    }
  }
  void record(hipEvent_t e, hipStream_t s) {
    // the stream's frontier: its last node and whatever it has waited for and not yet consumed
    std::vector<hipGraphNode_t> f;
    auto l = last.find(s);
    if (l != last.end() && l->second) f.push_back(l->second);
    auto p = pend.find(s);
    if (p != pend.end())
      for (hipGraphNode_t n : p->second) f.push_back(n);
    evnodes[e] = std::move(f);
  }
  void wait(hipStream_t s, hipEvent_t e) {
    auto it = evnodes.find(e);
    if (it == evnodes.end()) return;  // recorded before this graph began: already complete (the caller drained the streams)
    auto &p = pend[s];
    for (hipGraphNode_t n : it->second) p.push_back(n);
  }
};
inline thread_local GraphRec *tl_rec = nullptr;

// kernel launch: hipLaunchKernelGGL, or a kernel node of the graph under construction
template <class... KArgs, class... Args>
inline void klaunch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, Args &&...args) {
  if (!tl_rec) {
    hipLaunchKernelGGL(kernel, grid, block, (unsigned)shmem, s, static_cast<KArgs>(args)...);
    return;
  }
  std::tuple<KArgs...> held{static_cast<KArgs>(args)...};
  void *ptrs[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1];
  size_t i = 0;
  std::apply([&](auto &...a) { ((ptrs[i++] = (void *)&a), ...); }, held);
  hipKernelNodeParams p = {};
  p.func = (void *)kernel;
  p.gridDim = grid;
  p.blockDim = block;
  p.sharedMemBytes = (unsigned)shmem;
  p.kernelParams = ptrs;
  p.extra = nullptr;
  std::vector<hipGraphNode_t> d = tl_rec->deps(s);
  hipGraphNode_t n = nullptr;
  const hipError_t e = hipGraphAddKernelNode(&n, tl_rec->graph, d.data(), d.size(), &p);
  tl_rec->added(s, n, e);
}
#define GOGP_KLAUNCH(kernel, grid, block, shmem, stream, ...) \
  ::gogp::klaunch(kernel, grid, block, shmem, stream, ##__VA_ARGS__)

inline hipError_t rec_memcpy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
  if (!tl_rec) return hipMemcpyAsync(dst, src, bytes, kind, s);
  std::vector<hipGraphNode_t> d = tl_rec->deps(s);
  hipGraphNode_t n = nullptr;
  const hipError_t e = hipGraphAddMemcpyNode1D(&n, tl_rec->graph, d.data(), d.size(), dst, src, bytes, kind);
  tl_rec->added(s, n, e);
  return e;
}
inline hipError_t rec_memset_async(void *dst, int value, size_t bytes, hipStream_t s) {
  if (!tl_rec) return hipMemsetAsync(dst, value, bytes, s);
  hipMemsetParams mp = {};
  mp.dst = dst;
  mp.value = (unsigned)value & 0xffu;
  mp.elementSize = 1;
  mp.width = bytes;
  mp.height = 1;
  mp.pitch = bytes;
  std::vector<hipGraphNode_t> d = tl_rec->deps(s);
  hipGraphNode_t n = nullptr;
  const hipError_t e = hipGraphAddMemsetNode(&n, tl_rec->graph, d.data(), d.size(), &mp);
  tl_rec->added(s, n, e);
  return e;
}
inline hipError_t rec_event_record(hipEvent_t e, hipStream_t s) {
  if (!tl_rec) return hipEventRecord(e, s);
  tl_rec->record(e, s);
  return hipSuccess;
}
inline hipError_t rec_stream_wait(hipStream_t s, hipEvent_t e) {
  if (!tl_rec) return hipStreamWaitEvent(s, e, 0);
  tl_rec->wait(s, e);
  return hipSuccess;
}

}  // namespace gogp
