// Internals of the peer group (csrc/peer.hip) that the joint fit's fused reduction + exchange kernel needs
// (csrc/joint_fit.hip): the argument block of the exchange, the layout of an exchange region, the group object.
#pragma once
#include "lc_common.h"

namespace lc_peer {

constexpr int kMaxPeers = 16;
constexpr int kChunk = 1024;  // floats per chunk: one flag each (256 threads x 4 in the stand-alone kernel)

struct PeerArgs {
  int rank, world, count, cpad, nchunks;
  unsigned int seq;
  float *buf;               // the block, reduced in place
  float *xch[kMaxPeers];    // exchange region of every rank: [2][cpad] floats, then [2][nchunks] flags
  unsigned int *err;        // local: a wait ran out
};

__device__ __forceinline__ unsigned int *peer_flags(float *base, int cpad) { return (unsigned int *)(base + 2 * (size_t)cpad); }

// The consumer side of one chunk's hand-off, by ONE lane: wait (bounded by the constant 100 MHz counter: 2 s) until every peer
// has raised the chunk's flag of this call, then one system-scope acquire.  Returns 0 when a wait ran out (reported in *err;
// once set, every later call gives up at once: a broken exchange costs one time-out, not one per call).
__device__ __forceinline__ int peer_wait_chunk(const PeerArgs &A, int c, int par) {
  const unsigned int want = A.seq + 1u;
  int good = (__hip_atomic_load(A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) ? 1 : 0;
  const long long t0 = wall_clock64();
  for (int r = 0; r < A.world && good; ++r) {
    if (r == A.rank) continue;
    const unsigned int *fl = peer_flags(A.xch[r], A.cpad) + par * A.nchunks + c;
    int spins = 0;
    while ((int)(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
      __builtin_amdgcn_s_sleep(16);
      if ((++spins & 63) == 0 && wall_clock64() - t0 > 200000000ll) {
        good = 0;
        __hip_atomic_store(A.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  if (good && A.world > 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return good;
}

}  // namespace lc_peer

struct lc_peer_group {
  lc_ctx *ctx = nullptr;
  int rank = 0, world = 1, count = 0, cpad = 0, nchunks = 0;
  unsigned int seq = 0;
  float *own = nullptr;                          // this rank's exchange region
  float *peer[lc_peer::kMaxPeers] = {};          // mapped regions (peer[rank] == own)
  bool opened[lc_peer::kMaxPeers] = {};
  unsigned int *err = nullptr;
  size_t bytes = 0;
  // fused reduction + exchange of the joint fit (joint_reduce_peer_kernel): arrivals per chunk (never reset: the last of a
  // chunk's blocks of fused call q sees (q + 1) * blocks_of_the_chunk) and the number of fused calls so far
  unsigned int *arrive = nullptr;
  unsigned int fcalls = 0;
};

// Fills the argument block of the next exchange call of the group and counts the call (the caller launches a kernel that
// follows the hand-off protocol of peer_allreduce_kernel on `stream`).  *fcall: index of this fused call.
int lc_peer_next_call(lc_peer_group *g, int count, lc_peer::PeerArgs *A, unsigned int **arrive, unsigned int *fcall);
