// The reduction over the epochs and the peer-memory all-reduce of a sharded joint fit in ONE launch (sharded loop of
// lc_joint_run_sharded with the library's peer group as its transport: two launches before, joint_reduce_kernel and
// peer_allreduce_kernel).
//
// A block sums its pixels over the local epochs exactly as joint_reduce_kernel does (16-pixel tiles, reduce_pixels16's order;
// four tiles per block, so that the whole grid is resident at once: N^2 / 64 + 1 blocks), publishes them in the rank's exchange
// region with system-scope write-through stores and drains them.  The exchange protocol's flag is per 1024-float chunk, and
// a chunk is now the work of sixteen blocks: each counts itself into the chunk's arrival word (device scope), and the block
// that completes the count raises the chunk's flag (system-scope release).  Every block then waits for the same chunk's flag
// of every peer (peer_wait_chunk: bounded), reads the peers' values of ITS pixels with system-scope loads and adds in rank
// order - the sums of peer_allreduce_kernel, bit for bit (tests/test_distributed_gpu.py: two ranks on one GPU against the
// unsharded fit and against the host-staged collective).  The scalar block ([dc | flux moments | chi2 | epochs], the last
// chunk) does the same with reduce_scalars.
//
// Why no block can wait for ever: every block publishes and counts itself BEFORE it waits, so a rank's flags depend only on
// its blocks having started - and all N^2 / 64 + 1 of them are resident together (at most 1025 blocks of 256 threads, few
// registers: the host checks the occupancy) - never on a peer.  A peer that does not show up at all is the time-out of
// peer_wait_chunk, reported through the group's error word like in the two-launch form.
#pragma once
#include "joint_kernels.h"
#include "peer_shared.h"

namespace lc {

constexpr int kRpTiles = 4;   // 16-pixel tiles per block

__global__ __launch_bounds__(kRedThreads) void joint_reduce_peer_kernel(int E, int M, int NN, int need_h, const float *HG,
                                                                         const float *g_cx_e, const float *g_cy_e,
                                                                         const float *chi2_e, const float *a, const float *a_ref,
                                                                         float *shared, lc_peer::PeerArgs P, unsigned int *arrive,
                                                                         unsigned int fcall) {
  __shared__ float4 part[kRedParts][kRedPix / 4];
  __shared__ double lanes[kRedThreads];
  __shared__ int ok;
  const int nimg = NN / (kRedPix * kRpTiles);
  const int tid = threadIdx.x, b = blockIdx.x, par = (int)(P.seq & 1u);
  const bool img = b < nimg;
  float v[kRpTiles];
  int idx[kRpTiles], nval = 0;
  if (img) {
    float4 acc[kRpTiles];
#pragma unroll
    for (int tl = 0; tl < kRpTiles; ++tl)   // (the loads of all tiles in flight before the first is combined)
      acc[tl] = need_h ? reduce_pixels16_partial(E, NN, HG, (b * kRpTiles + tl) * kRedPix, tid) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tl = 0; tl < kRpTiles; ++tl) {
      if (tl > 0) __syncthreads();
      const float t = reduce_pixels16_combine(acc[tl], part, tid);
      v[tl] = need_h ? t : 0.f;
      idx[tl] = (b * kRpTiles + tl) * kRedPix + tid;
    }
    nval = (tid < kRedPix) ? kRpTiles : 0;
  } else {
    reduce_scalars(E, M, NN, g_cx_e, g_cy_e, chi2_e, a, a_ref, shared, lanes, tid);
    __syncthreads();   // (the values went to `shared` through this block's own stores)
#pragma unroll
    for (int tl = 0; tl < kRpTiles; ++tl) {
      v[tl] = 0.f;
      idx[tl] = 0;
    }
    if (tid < 4 * M + 2) {
      idx[0] = NN + tid;
      v[0] = shared[NN + tid];
      nval = 1;
    }
  }
  // chunk of this block's elements (64 consecutive pixels never straddle a 1024-float chunk; the scalars are the last chunk)
  const int c = img ? (b * kRpTiles * kRedPix) / lc_peer::kChunk : NN / lc_peer::kChunk;
  const unsigned int per_chunk = img ? (unsigned int)(lc_peer::kChunk / (kRpTiles * kRedPix)) : 1u;
  float *own = P.xch[P.rank] + (size_t)par * P.cpad;
#pragma unroll
  for (int tl = 0; tl < kRpTiles; ++tl)
    if (tl < nval) __hip_atomic_store(own + idx[tl], v[tl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned int seen = __hip_atomic_fetch_add(arrive + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen + 1u == (fcall + 1u) * per_chunk)   // this block completes the chunk: its flag, as the stand-alone kernel raises it
      __hip_atomic_store(lc_peer::peer_flags(P.xch[P.rank], P.cpad) + par * P.nchunks + c, P.seq + 1u, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    ok = lc_peer::peer_wait_chunk(P, c, par);
  }
  __syncthreads();
  if (!ok) {  // (the block keeps its local values, as in the two-launch form; the host reports the time-out)
#pragma unroll
    for (int tl = 0; tl < kRpTiles; ++tl)
      if (tl < nval && img) shared[idx[tl]] = v[tl];
    return;
  }
#pragma unroll
  for (int tl = 0; tl < kRpTiles; ++tl) {
    if (tl < nval) {
      float acc = 0.f;
      for (int r = 0; r < P.world; ++r) {  // rank order on every rank: identical sums everywhere
        float x = v[tl];
        if (r != P.rank)
          x = __hip_atomic_load(P.xch[r] + (size_t)par * P.cpad + idx[tl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        acc += x;
      }
      shared[idx[tl]] = acc;
    }
  }
}

}  // namespace lc
