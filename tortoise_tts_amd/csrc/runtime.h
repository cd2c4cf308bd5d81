// Host-side runtime pieces shared by the stage engines: device arena, the stream bridge that lets
// an engine run (and hipGraph-capture) on its own stream while staying ordered with the caller's,
// and a thin GEMM call builder.
#pragma once
#include <vector>
#include <stdlib.h>
#include "ops.h"

namespace tt {

// All engine workspaces come from hipMalloc at create time; nothing is allocated on the hot path.
struct Arena {
  std::vector<void*> ptrs;
  size_t total = 0;
  int alloc(void** out, size_t bytes, bool zero = true) {
    void* p = nullptr;
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      set_error("hipMalloc(%zu bytes) failed: %s (arena holds %zu bytes)", bytes, hipGetErrorString(e), total);
      return -2;
    }
    if (zero) {
      e = hipMemset(p, 0, bytes);
      if (e != hipSuccess) {
        set_error("hipMemset failed: %s", hipGetErrorString(e));
        return -2;
      }
    }
    ptrs.push_back(p);
    total += bytes;
    *out = p;
    return 0;
  }
  template <typename P> int alloc_t(P** out, size_t count, bool zero = true) { return alloc((void**)out, count * sizeof(P), zero); }
  void release() {
    for (void* p : ptrs) (void)hipFree(p);
    ptrs.clear();
    total = 0;
  }
};

// The caller hands us any stream (often the legacy null stream, which cannot be captured).  Work is
// enqueued on the engine's own stream between enter() and leave(), which order it after everything
// already queued on the caller's stream and make the caller's stream wait for it.
struct StreamBridge {
  hipStream_t own = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  int init() {
    TT_CHECK_HIP(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
    TT_CHECK_HIP(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
    TT_CHECK_HIP(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
    return 0;
  }
  int enter(hipStream_t user) {
    TT_CHECK_HIP(hipEventRecord(ev_in, user));
    TT_CHECK_HIP(hipStreamWaitEvent(own, ev_in, 0));
    return 0;
  }
  int leave(hipStream_t user) {
    TT_CHECK_HIP(hipEventRecord(ev_out, own));
    TT_CHECK_HIP(hipStreamWaitEvent(user, ev_out, 0));
    return 0;
  }
  void destroy() {
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
    if (own) (void)hipStreamDestroy(own);
    own = nullptr;
    ev_in = ev_out = nullptr;
  }
};

extern bool g_graph_replay;  // tt_graph_replay (common.hip): diagnostics switch, default on
static inline bool graphs_enabled() { return g_graph_replay; }

static inline GemmArgs gemm_args(const void* A, int lda, const void* W, int ldw, int M, int N, int K) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
  g.taps = 1; g.seq_len = M; g.splitk = 1; g.q_scale = 1.f;
  return g;
}

static inline int elem_size(int dtype) { return dtype_bytes(dtype); }
// p + elems operand elements of `es` bytes (2: bf16 / fp16, 4: the fp32 verification mode)
static inline void* offset_t(void* p, size_t elems, int es = 2) { return (void*)((char*)p + elems * es); }
static inline const void* offset_t(const void* p, size_t elems, int es = 2) { return (const void*)((const char*)p + elems * es); }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace tt
