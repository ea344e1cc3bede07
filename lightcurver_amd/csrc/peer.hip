// One-shot peer-memory all-reduce of the shared block of an epoch-sharded joint fit (include/lcmi.h, "peer group").
//
// The block [dL/dh (N^2) | dL/dc | flux moments | chi2 | n_epochs] is 64 - 256 KiB and is exchanged once per optimiser
// iteration (the reference has no counterpart - it keeps all epochs on one device,
// lightcurver/processes/roi_modelling.py:154-160,213).  At that size a ring collective is all latency (its launch and its
// 2 (N - 1) hops are of the order of the iteration itself); over xGMI every GPU reaches every other in one hop, so each
// rank publishes its block in an exchange buffer the others map through HIP IPC and every rank reads the N - 1 peers
// directly: one kernel, one hand-off, the sum taken in rank order by every rank (identical bits on all of them).
//
// Hand-off (per 4 KiB chunk, so no grid-wide step): the block's own values go to the exchange buffer with system-scope
// (write-through) stores, every storing wave drains them, workgroup barrier, one lane raises the chunk's flag with a
// system-scope release store; the lane then polls the same chunk's flag of every peer (bounded), barrier, and all lanes
// read the peers' values with system-scope loads (never served from a local cache).  Buffers and flags alternate between
// two parities: a rank can only be one call ahead of the slowest peer - it waits for that peer's flag of the current call,
// which the peer raises after it has finished reading the previous one - so a buffer is never rewritten while it is read.
#include <cstring>

#include "peer_shared.h"

using namespace lc_peer;

namespace {

__global__ __launch_bounds__(256) void peer_allreduce_kernel(PeerArgs A) {
  const int c = blockIdx.x, tid = threadIdx.x, par = (int)(A.seq & 1u);
  const int i0 = c * kChunk + tid * 4;
  float mine[4];
  float *own = A.xch[A.rank] + (size_t)par * A.cpad;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mine[k] = (i0 + k < A.count) ? A.buf[i0 + k] : 0.f;
    if (i0 + k < A.count) __hip_atomic_store(own + i0 + k, mine[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok;
  if (tid == 0) {
    const unsigned int want = A.seq + 1u;
    __hip_atomic_store(peer_flags(A.xch[A.rank], A.cpad) + par * A.nchunks + c, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // exit condition every workgroup reaches: a peer that never shows up is reported, not waited for - bounded by the
    // constant 100 MHz counter (s_memrealtime; 2 s), not by a spin count whose duration depends on the clock and on the
    // latency of the xGMI hop.  Then one system-scope acquire by the polling lane, behind the last successful poll and in front
    // of the workgroup barrier that releases the readers (the consumer side of the release the peers' flag stores carry).
    // The readers' loads are system-scope atomics - never served from a cache of this device - so the fence orders rather
    // than invalidates for them; it is one buffer_inv per 4 KiB chunk, 16 - 64 of them per call (measured at world size 1:
    // no change of the sharded loop's 90 us per iteration; an agent-scope acquire by every block of the 1 000-block update
    // cost 33 us).  (peer_wait_chunk, peer_shared.h)
    const int good = peer_wait_chunk(A, c, par);
    ok = good;
  }
  __syncthreads();
  if (!ok) return;  // the block keeps its local values; the host reports the time-out
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < A.world; ++r) {  // rank order on every rank: identical sums everywhere
    const float *src = A.xch[r] + (size_t)par * A.cpad;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = mine[k];
      if (r != A.rank && i0 + k < A.count) v = __hip_atomic_load(src + i0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      acc[k] += v;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (i0 + k < A.count) A.buf[i0 + k] = acc[k];
}

}  // namespace

extern "C" {

int lc_peer_group_create(lc_ctx *ctx, int count, int rank, int world, lc_peer_group **out) {
  if (!ctx || !out || count <= 0 || world < 1 || world > kMaxPeers || rank < 0 || rank >= world) {
    if (ctx) ctx->err = "lc_peer_group_create: invalid argument (at most 16 ranks)";
    return LC_ERR_INVALID;
  }
  LC_ENTER(ctx);
  lc_peer_group *g = new lc_peer_group();
  g->ctx = ctx;
  g->rank = rank;
  g->world = world;
  g->count = count;
  g->nchunks = (count + kChunk - 1) / kChunk;
  g->cpad = g->nchunks * kChunk;
  g->bytes = (2 * (size_t)g->cpad + 2 * (size_t)g->nchunks) * sizeof(float);
  // fine-grained (uncached across devices) where the runtime offers it; plain device memory otherwise
  hipError_t e = hipExtMallocWithFlags((void **)&g->own, g->bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc((void **)&g->own, g->bytes);
  }
  if (e != hipSuccess || hipMalloc((void **)&g->err, sizeof(unsigned int)) != hipSuccess) {
    ctx->err = std::string("lc_peer_group_create: ") + hipGetErrorString(e);
    if (g->own) (void)hipFree(g->own);
    delete g;
    return LC_ERR_DEVICE;
  }
  LC_HIP(ctx, hipMemset(g->own, 0, g->bytes));
  LC_HIP(ctx, hipMemset(g->err, 0, sizeof(unsigned int)));
  LC_HIP(ctx, hipMalloc((void **)&g->arrive, (size_t)g->nchunks * sizeof(unsigned int)));
  LC_HIP(ctx, hipMemset(g->arrive, 0, (size_t)g->nchunks * sizeof(unsigned int)));
  g->peer[rank] = g->own;
  *out = g;
  return LC_OK;
}

int lc_peer_group_export(lc_peer_group *g, void *handle_out, int handle_bytes) {
  if (!g || !handle_out || handle_bytes < (int)sizeof(hipIpcMemHandle_t)) return LC_ERR_INVALID;
  LC_ENTER(g->ctx);
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, g->own);
  if (e != hipSuccess && g->seq == 0) {
    // a runtime that does not export the fine-grained allocation: the same region as plain device memory (the kernel's
    // accesses are system-scope either way)
    (void)hipGetLastError();
    float *plain = nullptr;
    if (hipMalloc((void **)&plain, g->bytes) == hipSuccess && hipMemset(plain, 0, g->bytes) == hipSuccess) {
      (void)hipFree(g->own);
      g->own = g->peer[g->rank] = plain;
      e = hipIpcGetMemHandle(&h, g->own);
    } else if (plain) {
      (void)hipFree(plain);
    }
  }
  LC_HIP(g->ctx, e);
  std::memset(handle_out, 0, (size_t)handle_bytes);
  std::memcpy(handle_out, &h, sizeof(h));
  return LC_OK;
}

int lc_peer_group_connect(lc_peer_group *g, const void *handles, int handle_bytes) {
  if (!g || !handles || handle_bytes < (int)sizeof(hipIpcMemHandle_t)) return LC_ERR_INVALID;
  LC_ENTER(g->ctx);
  for (int r = 0; r < g->world; ++r) {
    if (r == g->rank || g->opened[r]) continue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, (const char *)handles + (size_t)r * handle_bytes, sizeof(h));
    void *p = nullptr;
    LC_HIP(g->ctx, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    g->peer[r] = (float *)p;
    g->opened[r] = true;
  }
  return LC_OK;
}

// matches lc_allreduce_fn (user = the group): the callback of lc_joint_run_sharded, or called directly
int lc_peer_allreduce(void *user, void *dev_buf, int count, void *hip_stream) {
  lc_peer_group *g = (lc_peer_group *)user;
  if (!g || !dev_buf) return LC_ERR_INVALID;
  LC_ENTER(g->ctx);
  if (count != g->count) LC_FAIL(g->ctx, LC_ERR_INVALID, "lc_peer_allreduce: the group was created for another block length");
  for (int r = 0; r < g->world; ++r)
    if (!g->peer[r]) LC_FAIL(g->ctx, LC_ERR_INVALID, "lc_peer_allreduce: lc_peer_group_connect has not mapped every peer");
  PeerArgs A;
  std::memset(&A, 0, sizeof(A));
  A.rank = g->rank;
  A.world = g->world;
  A.count = g->count;
  A.cpad = g->cpad;
  A.nchunks = g->nchunks;
  A.seq = g->seq;
  A.buf = (float *)dev_buf;
  for (int r = 0; r < g->world; ++r) A.xch[r] = g->peer[r];
  A.err = g->err;
  hipLaunchKernelGGL(peer_allreduce_kernel, dim3(g->nchunks), dim3(256), 0, (hipStream_t)hip_stream, A);
  LC_HIP(g->ctx, hipGetLastError());
  g->seq += 1;
  return LC_OK;
}

// 0 = every wait so far was answered; LC_ERR_DEVICE = a peer did not show up in time (synchronises the context's stream)
int lc_peer_group_status(lc_peer_group *g) {
  if (!g) return LC_ERR_INVALID;
  LC_ENTER(g->ctx);
  LC_HIP(g->ctx, hipStreamSynchronize(g->ctx->stream));
  unsigned int e = 0;
  LC_HIP(g->ctx, hipMemcpy(&e, g->err, sizeof(e), hipMemcpyDeviceToHost));
  if (e) LC_FAIL(g->ctx, LC_ERR_DEVICE, "peer all-reduce: a rank did not publish its block in time");
  return LC_OK;
}

void lc_peer_group_destroy(lc_peer_group *g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  for (int r = 0; r < g->world; ++r)
    if (g->opened[r]) (void)hipIpcCloseMemHandle(g->peer[r]);
  if (g->own) (void)hipFree(g->own);
  if (g->err) (void)hipFree(g->err);
  if (g->arrive) (void)hipFree(g->arrive);
  delete g;
}

}  // extern "C"

int lc_peer_next_call(lc_peer_group *g, int count, lc_peer::PeerArgs *A, unsigned int **arrive, unsigned int *fcall) {
  if (!g || !A || !arrive || !fcall) return LC_ERR_INVALID;
  LC_ENTER(g->ctx);
  if (count != g->count) LC_FAIL(g->ctx, LC_ERR_INVALID, "peer exchange: the group was created for another block length");
  for (int r = 0; r < g->world; ++r)
    if (!g->peer[r]) LC_FAIL(g->ctx, LC_ERR_INVALID, "peer exchange: lc_peer_group_connect has not mapped every peer");
  std::memset(A, 0, sizeof(*A));
  A->rank = g->rank;
  A->world = g->world;
  A->count = g->count;
  A->cpad = g->cpad;
  A->nchunks = g->nchunks;
  A->seq = g->seq;
  for (int r = 0; r < g->world; ++r) A->xch[r] = g->peer[r];
  A->err = g->err;
  *arrive = g->arrive;
  *fcall = g->fcalls;
  g->seq += 1;
  g->fcalls += 1;
  return LC_OK;
}
