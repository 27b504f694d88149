// fake_rccl.cpp - TEST INFRASTRUCTURE: a stand-in collective library with the four NCCL entry points libdust_amd binds (ncclGetUniqueId,
// ncclCommInitRank, ncclAllGather, ncclCommDestroy), for ranks that are PROCESSES SHARING ONE GPU.  The real RCCL wants one GPU per
// rank; the build container's GPU box has one, so the C-side sharded tick (dust_amd.hip sharded_steps / sharded_forward: collective
// order, in-place offsets, the particle all-gather on the side stream under the next iteration's rollouts) had never run at world > 1
// (VERDICT r4).  Selected with DUST_RCCL_LIB=<this library>.
//
// ncclAllGather(send, recv, count, type, comm, stream), in place (send == recv + rank * count), synchronously:
//   1. hipStreamSynchronize(stream) - ONLY the stream the caller passed: if the library forgot to order that stream behind the kernels
//      that produce the rank's piece (they run on another stream), the peers read stale data and the test fails - the point of the test;
//   2. the rank publishes an IPC handle of the allocation that holds `recv` (hipIpcGetMemHandle) and its offset in a POSIX
//      shared-memory block; barrier;
//   3. it opens every peer's handle and copies that peer's piece into its own buffer; barrier (nobody overwrites what a peer still reads).
// A legal, if slow, implementation of the call's contract.  Build: g++ -shared -fPIC fake_rccl.cpp -I/opt/rocm/include -L/opt/rocm/lib -lamdhip64 -lrt
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>

namespace {
enum { MAX_RANKS = 8 };
struct Slot {
  hipIpcMemHandle_t handle;
  unsigned long long offset;
  unsigned long long base_id;  // changes when the allocation behind the buffer changes
};
struct Shared {
  std::atomic<int> attached;
  std::atomic<unsigned int> barrier_count;
  std::atomic<unsigned int> barrier_gen;
  std::atomic<int> failed;
  Slot slot[MAX_RANKS];
};
struct Comm {
  int rank, world;
  Shared *sh;
  std::string name;
  std::map<std::string, void *> opened;  // peer handle bytes -> mapped base
  unsigned long long calls;
};
struct UniqueId {
  char internal[128];
};

bool barrier(Comm *c) {
  Shared *s = c->sh;
  const unsigned int gen = s->barrier_gen.load();
  if (s->barrier_count.fetch_add(1) + 1 == (unsigned int)c->world) {
    s->barrier_count.store(0);
    s->barrier_gen.fetch_add(1);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (s->barrier_gen.load() == gen) {
    if (s->failed.load()) return false;
    std::this_thread::yield();
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
      s->failed.store(1);
      return false;
    }
  }
  return true;
}
}  // namespace

extern "C" {

int ncclGetVersion(int *v) {
  *v = 22606;  // an NCCL 2.x version code
  return 0;
}

const char *ncclGetErrorString(int) { return "fake_rccl: collective failed (a peer did not arrive within 60 s, or a HIP call failed)"; }

int ncclGetUniqueId(UniqueId *id) {
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/dust_fake_rccl_%d_%lld", (int)getpid(),
           (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return 0;
}

int ncclCommInitRank(void **out, int nranks, UniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return 1;
  int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return 1;
  if (ftruncate(fd, sizeof(Shared)) != 0) return 1;
  void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 1;
  Comm *c = new Comm();
  c->rank = rank;
  c->world = nranks;
  c->sh = reinterpret_cast<Shared *>(p);  // (a fresh shm object is zero-filled: all atomics start at 0)
  c->name = id.internal;
  c->calls = 0;
  c->sh->attached.fetch_add(1);
  const auto t0 = std::chrono::steady_clock::now();
  while (c->sh->attached.load() < nranks) {  // ncclCommInitRank is a collective
    std::this_thread::yield();
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return 1;
  }
  *out = c;
  return 0;
}

int ncclCommDestroy(void *comm) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (!c) return 0;
  for (auto &kv : c->opened) (void)hipIpcCloseMemHandle(kv.second);
  if (c->rank == 0) shm_unlink(c->name.c_str());
  munmap(c->sh, sizeof(Shared));
  delete c;
  return 0;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (!c || dtype != 7) return 1;  // ncclFloat32
  const size_t bytes = count * 4;
  if (send != static_cast<const char *>(recv) + (size_t)c->rank * bytes) return 1;  // in place only
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (c->world == 1) return 0;
  void *base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t *>(&base), &size, reinterpret_cast<hipDeviceptr_t>(recv)) != hipSuccess) return 1;
  Slot &mine = c->sh->slot[c->rank];
  if (hipIpcGetMemHandle(&mine.handle, base) != hipSuccess) return 1;
  mine.offset = (unsigned long long)(static_cast<char *>(recv) - static_cast<char *>(base));
  mine.base_id = (unsigned long long)(uintptr_t)base;
  if (!barrier(c)) return 1;
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    const Slot &s = c->sh->slot[r];
    const std::string key(reinterpret_cast<const char *>(&s.handle), sizeof s.handle);
    auto it = c->opened.find(key);
    void *peer = nullptr;
    if (it == c->opened.end()) {
      if (hipIpcOpenMemHandle(&peer, s.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        c->sh->failed.store(1);
        return 1;
      }
      c->opened[key] = peer;
    } else {
      peer = it->second;
    }
    const char *src = static_cast<const char *>(peer) + s.offset + (size_t)r * bytes;
    // (on the caller's stream, like the real collective: what runs on OTHER streams of the process is not ordered by this call)
    if (hipMemcpyAsync(static_cast<char *>(recv) + (size_t)r * bytes, src, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) {
      c->sh->failed.store(1);
      return 1;
    }
  }
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;  // (the peers may overwrite their pieces once everybody has copied)
  if (!barrier(c)) return 1;
  c->calls++;
  return 0;
}
}
