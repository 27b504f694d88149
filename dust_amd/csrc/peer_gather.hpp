// peer_gather.hpp - the sharded tick's in-place all-gathers as DIRECT PEER STORES (round 6; SURVEY 8e: "an RCCL all-gather over xGMI
// of particle states before the pairwise kernel step"; svmpc.py:38-83 is what a rank runs on its rows between them).
//
// The three exchanges of a sharded tick move little data - a rank's score / particle rows (655 KB at cfg4 over 8 ranks) and its N / G
// log-weights - between GPUs that are ONE xGMI hop apart (a fully connected mesh: 7 links per GPU).  A ring all-gather passes every
// piece over G - 1 hops behind G - 1 flag hand-shakes; here every rank writes its piece straight into every peer's buffer (the peers'
// buffers are mapped through HIP IPC once, when the communicator is set up) and raises one arrival word per peer:
//
//   peer_store_kernel   blockIdx.y = peer: copy the rank's piece to the same offset of that peer's buffer (16-byte stores over the
//                       link), system-scope fence; the last workgroup of a peer's column writes  flags[peer][which][rank] = seq
//   peer_wait_kernel    one wave: lane r spins until  flags[me][which][r] >= seq  (system-scope loads; bounded - a rank that died
//                       raises the error word instead of hanging the stream).  The kernels behind it in the stream start after the
//                       arrival of every piece, and a kernel boundary is where a GPU's caches are made coherent with what peers
//                       wrote into its memory - the mechanism the collective library's own kernels rely on.
//
// Write-after-read.  A peer's next store into this rank's SCORE or LOG-WEIGHT buffer comes after that peer has waited for this rank's
// later particle piece, which this rank sends only after the kernels that read the earlier pieces (score -> Stein pass -> update ->
// particles -> prior pass / log p -> log-weights -> finalize + roll -> next tick's score ...): no hand-shake needed.  The PARTICLE
// buffer is different: a peer that has this rank's score rows can finish its Stein pass and update and store its new particles
// while this rank's own Stein pass is still reading the old ones (the collective library's all-gather cannot run ahead like that: it
// needs this rank's call).  So a particle store to peer g starts with a token exchange: the column of workgroups that serves g
// first tells g "my readers of your old rows are done" (they are: the store kernel sits behind them in the stream) and then waits
// for g's token before it writes.  Both tokens are sent unconditionally at the head of the two kernels: no cycle.
// Sequence numbers count up per buffer; words are never reset.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dust {

enum { PEER_MAX = 8, PEER_BUFS = 3, PEER_ROWS = 4 };  // (score, particles, log-weights); flag rows: their arrival words + the particle tokens

struct PeerStoreArgs {
  const float *src;      // the rank's piece (already at its offset of the rank's own buffer)
  size_t count;          // floats of the piece
  size_t offset;         // its offset (floats) in every buffer
  float *dst[PEER_MAX];  // the peers' buffers (entry `rank`: unused)
  unsigned int *flags[PEER_MAX];  // every rank's flag block [PEER_ROWS][PEER_MAX] (entry `rank`: this rank's own)
  int handshake;         // exchange tokens with the peer before writing (the particle buffer)
  unsigned long long timeout_ticks;
  unsigned int *done;    // [PEER_MAX] workgroup counters of this rank (zero between launches)
  int world, rank, which;
  unsigned int seq;
};

__global__ __launch_bounds__(256) void peer_store_kernel(const PeerStoreArgs a) {
  const int p = blockIdx.y, g = p < a.rank ? p : p + 1;  // the peer this column of workgroups writes to
  float *dst = a.dst[g] + a.offset;
  if (a.handshake) {
    if (threadIdx.x == 0) {
      if (blockIdx.x == 0) __hip_atomic_store(a.flags[g] + PEER_BUFS * PEER_MAX + a.rank, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned int *tok = a.flags[a.rank] + PEER_BUFS * PEER_MAX + g;
      const unsigned long long t0 = wall_clock64();
      while ((int)(__hip_atomic_load(tok, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - a.seq) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > a.timeout_ticks) break;  // (the peer's wait kernel will report the piece that never arrives)
      }
    }
    __syncthreads();
  }
  // 16-byte WRITE-THROUGH stores (sc0 sc1: performed at system scope - the peer's memory - not parked in this GPU's L2), so that the
  // wave only has to wait for its own stores (vmcnt) instead of writing the whole L2 back with a system-scope fence per wave
  // (measured on one device at 8 ranks: 90 us per tick for the three exchanges with the fence, the L2 full of the tick's dirty lines)
  const size_t n4 = a.count >> 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    typedef float pg_v4f __attribute__((ext_vector_type(4)));
    const pg_v4f v = reinterpret_cast<const pg_v4f *>(a.src)[i];
    // (s_nop behind the store: a > 8-byte VMEM store reads its data registers late - stein.hpp store16)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(reinterpret_cast<pg_v4f *>(dst) + i), "v"(v) : "memory");
  }
  if (blockIdx.x == 0)
    for (size_t i = (n4 << 2) + threadIdx.x; i < a.count; i += blockDim.x)
      asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(dst + i), "v"(a.src[i]) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are performed at the peer
  __shared__ unsigned int last;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(a.done + g, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = prev + 1u == gridDim.x ? 1u : 0u;
    if (last) {
      __hip_atomic_store(a.done + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.flags[g] + a.which * PEER_MAX + a.rank, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

struct PeerWaitArgs {
  const unsigned int *flags;  // this rank's arrival words [PEER_BUFS][PEER_MAX]
  unsigned int *err;          // raised (1) when a piece did not arrive within the bound
  int world, rank, which;
  unsigned int seq;
  unsigned long long timeout_ticks;  // wall_clock64 ticks (100 MHz)
};

__global__ __launch_bounds__(64) void peer_wait_kernel(const PeerWaitArgs a) {
  const int r = threadIdx.x;
  if (r >= a.world || r == a.rank) return;
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    const unsigned int v = __hip_atomic_load(a.flags + a.which * PEER_MAX + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)(v - a.seq) >= 0) return;
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > a.timeout_ticks) {
      __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}

}  // namespace dust
