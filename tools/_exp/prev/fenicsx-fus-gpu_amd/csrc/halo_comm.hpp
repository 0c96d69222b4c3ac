// Ghost-dof halo exchange behind the C ABI (fus_halo_{create,forward,reverse,destroy}).
//
// Replaces the closures of  cuda/scatterer.py:104-188 (scatter_reverse) and :191-277
// (scatter_forward), which per neighbour launch one pack kernel, device-synchronise, post
// MPI Isend/Irecv on device pointers, wait, launch one unpack kernel and synchronise again, and the
// C++ driver's scatter calls (cpp/common/Linear.hpp:120,193,196,212).
//
// MI355X form: the library owns one HIGH-PRIORITY stream per communicator; an exchange is
//   [caller's stream: event "vector ready"]
//   comm stream: wait -> pack (ONE launch for all neighbours) -> ncclGroupStart; ncclSend / ncclRecv per
//                neighbour; ncclGroupEnd (RCCL over xGMI: a neighbour all-to-all-v, no host sync) ->
//                unpack (one launch) -> event "done"
//   [caller's stream: wait "done"]                                     <- fus_halo_*_end
// so between begin and end the caller's stream is free for interior-cell kernels, and not even the
// pack / unpack launches sit between them.  When a rank's ghosts are numbered owner by owner (the
// ghost block of a vector IS the concatenation of the owners' messages) the forward exchange
// receives straight into the vector and the reverse exchange sends straight from it: no
// unpack_fwd / pack_rev launch at all.
//
// Transports:
//   RCCL   librccl.so.1 resolved with dlopen at first use (libfusgpu.so itself has no link-time
//          dependency on it: the operator kernels load on any ROCm box).  Bootstrap = 128-byte unique
//          id from rank 0, broadcast by whatever the host already has (MPI_Bcast in the reference's
//          drivers, torch.distributed here); one process per GPU.
//   LOCAL  all ranks live in ONE process (tests on a one-GPU box, or one process driving several
//          GPUs): the receiver pulls each message with hipMemcpyAsync on its comm stream, ordered
//          by events.  Host-side contract: every rank's *_begin of an exchange is called before any
//          rank's *_end of it.
//   PEER   halo_ipc.hpp: the neighbours' receive arenas are mapped once (HIP IPC handles exchanged by the
//          host's bootstrap channel); an exchange is a send kernel that stores straight into the neighbour's
//          arena and a receive kernel that waits on a sequence flag -- two kernels of a few registers that DO
//          run next to a chip-filling operator launch, where RCCL's 264-register kernel does not.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <strings.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "halo.hpp"
#include "halo_ipc.hpp"
#include "plan.hpp"

// The few RCCL declarations the dlopen'ed entry points need (rccl/rccl.h is not required to build the library).
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;    // ncclSuccess == 0
typedef int ncclDataType_t;  // ncclFloat32 == 7, ncclFloat64 == 8 (nccl.h, stable since NCCL 2.0)
}
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclFloat32 = 7, ncclFloat64 = 8;

namespace fus {

// ------------------------------------------------------------------------------------ RCCL, lazily
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;

  bool load() {
    if (handle) return true;
    // a process that already holds an RCCL (torch does) gets that one: same SONAME
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (handle) break;
    }
    if (!handle) {
      error = std::string("cannot load librccl.so.1: ") + dlerror();
      return false;
    }
#define FUS_SYM(field, sym)                                          \
  field = reinterpret_cast<decltype(field)>(dlsym(handle, sym));     \
  if (!field) {                                                      \
    error = std::string("librccl lacks ") + sym;                     \
    handle = nullptr;                                                \
    return false;                                                    \
  }
    FUS_SYM(GetUniqueId, "ncclGetUniqueId")
    FUS_SYM(CommInitRank, "ncclCommInitRank")
    FUS_SYM(CommDestroy, "ncclCommDestroy")
    FUS_SYM(GroupStart, "ncclGroupStart")
    FUS_SYM(GroupEnd, "ncclGroupEnd")
    FUS_SYM(Send, "ncclSend")
    FUS_SYM(Recv, "ncclRecv")
    FUS_SYM(GetErrorString, "ncclGetErrorString")
#undef FUS_SYM
    return true;
  }
};

inline RcclApi& rccl() {
  static RcclApi api;
  return api;
}

// ------------------------------------------------------------------------------------ communicator
struct Halo;

struct LocalWorld {  // LOCAL transport: the ranks of one process
  int nranks = 0;
  std::vector<std::vector<Halo*>> halos;  // [rank][creation index]
};

struct Comm {
  enum Kind { RCCL = 0, LOCAL = 1, PEER = 2 } kind = RCCL;
  int rank = 0, nranks = 1, device = 0;
  ncclComm_t nccl = nullptr;
  std::shared_ptr<LocalWorld> world;
  hipStream_t stream = nullptr;   // high priority: small exchange kernels between big operator kernels
  hipStream_t stream2 = nullptr;  // PEER: the receive kernels' stream (sends never queue behind a waiting receive)
  int nhalos = 0;                 // live halo objects: the communicator outlives them
  std::vector<Halo*> halos;       // the live halo objects (comm_health)
  // event-free fork / join between a caller's stream and ``stream`` (comm_fork / comm_join below)
  uint64_t* sync_words = nullptr;  // device: [0] fork flag, [1] join flag, [2..] status (ST_TIMEOUTS, ST_DEAD)
  uint64_t sync_seq[2] = {0, 0};
  uint64_t sync_budget = 0;
  // One sequence flag per direction: consecutive forks / joins of a communicator must come from ONE caller stream (two
  // streams forking alternately would let the later stream's signal satisfy the earlier wait).  Enforced: a fork from
  // another stream is accepted only when the communicator's stream has drained.
  hipStream_t caller_stream = nullptr;
  bool caller_stream_set = false;
  // PEER: fork / join folded into the exchange kernels (fus_comm_fork_lazy / fus_comm_arm_join)
  uint64_t gate_pending = 0;      // fork sequence number no kernel of the communicator's stream waits for yet
  bool join_armed = false;        // the last receive kernel of the next begin / begin_group publishes the join flag
  Halo* join_halo = nullptr;      // ... that is this halo's, in direction join_dir
  int join_dir = 0;
  uint64_t join_inflight = 0;     // join sequence number a posted receive kernel will publish
  std::string last_error;
};

inline std::mutex& local_worlds_mutex() {
  static std::mutex m;
  return m;
}

inline std::map<int, std::weak_ptr<LocalWorld>>& local_worlds() {  // guarded by local_worlds_mutex()
  static std::map<int, std::weak_ptr<LocalWorld>> m;
  return m;
}

inline hipError_t comm_make_stream(Comm* c) {
  int lo = 0, hi = 0;
  hipError_t e = hipGetDevice(&c->device);
  if (e != hipSuccess) return e;
  e = hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically lowest = highest priority
  if (e != hipSuccess) return e;
  if (const char* pr = std::getenv("FUS_COMM_PRIORITY")) {  // experiments: "normal" / "low" instead of the highest
    if (!std::strcmp(pr, "normal")) hi = 0;
    if (!std::strcmp(pr, "low")) hi = lo;
  }
  e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi);
  if (e == hipSuccess && c->kind == Comm::PEER) {
    // One stream by default: send, receive and -- when the host puts them there (fus_comm_stream) -- the boundary-cell
    // kernels between a forward and a reverse exchange follow each other in stream order, with no event edge between
    // them.  FUS_IPC_TWO_STREAMS=1: receive kernels on a stream of their own.
    const char* two = std::getenv("FUS_IPC_TWO_STREAMS");
    if (two && two[0] == '1')
      e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi);
    else
      c->stream2 = c->stream;
  }
  return e;
}
inline hipError_t comm_sync_init(Comm* c);
// stream(s) + the fork / join words, all at creation: the first fork of a time loop must not synchronise the device
inline hipError_t comm_make_stream_and_sync(Comm* c) {
  hipError_t e = comm_make_stream(c);
  if (e == hipSuccess) e = comm_sync_init(c);
  return e;
}

// ------------------------------------------------------------------------------------ fork / join without events
// Ordering a side stream after the caller's stream with an event costs the caller's stream 7 us per fork next to
// chip-filling launches (the record is a marker with a cache write-back between the caller's kernels), the join
// another 3-4 us (profiles/r03g_forkjoin.log).  A one-thread kernel in the caller's stream costs 2.4 us.  So:
//   fork   caller's stream: signal kernel (flag = seq: runs when everything before it in that stream has completed);
//          communicator's stream: a one-wave kernel that waits (bounded) for flag >= seq -- what follows it in that
//          stream starts after it, by stream order;
//   join   the same with the roles exchanged.
// Data written before the signal kernel is visible after the wait kernel for the reason two consecutive kernels of one
// stream see each other's data: every kernel ends with a release and starts with an acquire at device scope.
__global__ void stream_signal_kernel(uint64_t* flag, uint64_t seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void stream_wait_kernel(const uint64_t* flag, uint64_t seq, uint64_t* status, uint64_t budget) {
  if (threadIdx.x != 0) return;
  if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= seq) return;
  if (__hip_atomic_load(&status[ST_DEAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const uint64_t t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > budget) {
      atomicAdd((unsigned long long*)&status[ST_TIMEOUTS], 1ull);
      __hip_atomic_store(&status[ST_DEAD], (uint64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
}

inline hipError_t comm_sync_init(Comm* c) {
  if (c->sync_words) return hipSuccess;
  hipError_t e = hipMalloc(&c->sync_words, (2 + ST_WORDS) * sizeof(uint64_t));
  if (e == hipSuccess) e = hipMemset(c->sync_words, 0, (2 + ST_WORDS) * sizeof(uint64_t));
  int khz = 100000;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device);
  double seconds = 20.0;
  if (const char* v = std::getenv("FUS_IPC_SPIN_SECONDS")) seconds = std::atof(v) > 0 ? std::atof(v) : seconds;
  c->sync_budget = (uint64_t)(seconds * 1e3 * khz);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  return e;
}

// the wait of a lazily posted fork that no send kernel took over: a wait kernel on the communicator's stream after all
inline hipError_t comm_flush_gate(Comm* c) {
  if (!c->gate_pending) return hipSuccess;
  const uint64_t seq = c->gate_pending;
  c->gate_pending = 0;
  hipLaunchKernelGGL(stream_wait_kernel, dim3(1), dim3(64), 0, c->stream, c->sync_words + 0, seq, c->sync_words + 2, c->sync_budget);
  return hipGetLastError();
}

// a fork signal that was attached to "the next planned operator launch" and has not been carried by one: publish it with
// a signal kernel after all (no launch came: an empty cell range, a plan-free kernel)
inline hipError_t comm_flush_attached(Comm* c) {
  if (!c->sync_words) return hipSuccess;
  hipStream_t st = nullptr;
  const LaunchSignal s = take_launch_signal_of(c->sync_words + 0, &st);
  if (!s.flag) return hipSuccess;
  hipLaunchKernelGGL(stream_signal_kernel, dim3(1), dim3(1), 0, st, s.flag, s.seq);
  return hipGetLastError();
}

// which: 0 fork (``stream`` -> communicator's stream), 1 join (communicator's stream -> ``stream``)
// lazy (fork, PEER): no wait kernel; the first send kernel of the next exchange posted on the communicator's stream waits for
// the flag itself (halo_ipc_post) -- one kernel less in the exchange chain.
// attach (fork): no signal kernel; the NEXT PLANNED OPERATOR LAUNCH on ``stream`` publishes the flag when its first workgroup
// starts (plan.hpp LaunchSignal) -- 2.4 us less on the caller's stream.  Flushed by the next fork / join if no launch came.
// *misuse: the single-caller-stream contract was violated (nothing was launched; last_error says what).
inline hipError_t comm_fork_join(Comm* c, hipStream_t stream, int which, bool lazy, bool* misuse, bool attach = false) {
  *misuse = false;
  if (stream == c->stream) return hipSuccess;
  hipError_t e = comm_sync_init(c);
  if (e != hipSuccess) return e;
  if (c->caller_stream_set && stream != c->caller_stream) {
    // another caller stream: fine once everything forked so far has run (no wait kernel is pending on either side)
    if (which == 1 || hipStreamQuery(c->stream) != hipSuccess) {
      (void)hipGetLastError();
      c->last_error = which == 1 ? "fus_comm_join: called with a different stream than the fus_comm_fork before it"
                                 : "fus_comm_fork: consecutive forks of a communicator must come from one caller stream (the communicator's stream "
                                   "still has work forked from another stream; synchronise it before changing the caller stream)";
      *misuse = true;
      return hipSuccess;
    }
  }
  if (which == 0) {
    c->caller_stream = stream;
    c->caller_stream_set = true;
  }
  e = comm_flush_attached(c);
  if (e != hipSuccess) return e;
  e = comm_flush_gate(c);
  if (e != hipSuccess) return e;
  if (which == 1 && c->join_inflight) {  // a receive kernel already on the communicator's stream publishes the flag
    const uint64_t seq = c->join_inflight;
    c->join_inflight = 0;
    hipLaunchKernelGGL(stream_wait_kernel, dim3(1), dim3(64), 0, stream, c->sync_words + 1, seq, c->sync_words + 2, c->sync_budget);
    return hipGetLastError();
  }
  if (which == 1) {
    c->join_armed = false;  // armed, but no receive kernel took it (no neighbours on that side)
    c->join_halo = nullptr;
  }
  const uint64_t seq = ++c->sync_seq[which];
  hipStream_t from = which == 0 ? stream : c->stream, to = which == 0 ? c->stream : stream;
  if (which == 0 && attach) {
    const LaunchSignal old = post_launch_signal(stream, c->sync_words + 0, seq);
    if (old.flag) hipLaunchKernelGGL(stream_signal_kernel, dim3(1), dim3(1), 0, stream, old.flag, old.seq);  // another communicator's, same stream
  } else {
    hipLaunchKernelGGL(stream_signal_kernel, dim3(1), dim3(1), 0, from, c->sync_words + which, seq);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (which == 0 && lazy && c->kind == Comm::PEER && c->stream2 == c->stream) {
    c->gate_pending = seq;
    return hipSuccess;
  }
  hipLaunchKernelGGL(stream_wait_kernel, dim3(1), dim3(64), 0, to, c->sync_words + which, seq, c->sync_words + 2, c->sync_budget);
  return hipGetLastError();
}

// PEER: the last receive kernel of the NEXT fus_halo_*_begin / *_begin_group of this communicator also publishes the join
// flag, so that the fus_comm_join that follows launches only the wait kernel on the caller's stream.  That exchange must be
// the last thing enqueued on the communicator's stream before the join.  Without a receive kernel to carry it (no
// neighbours on that side, receive kernels on a stream of their own) the join falls back to its signal kernel.
inline hipError_t comm_arm_join(Comm* c) {
  hipError_t e = comm_sync_init(c);
  if (e != hipSuccess) return e;
  if (c->kind == Comm::PEER && c->stream2 == c->stream) {
    c->join_armed = true;
    c->join_halo = nullptr;
  }
  return hipSuccess;
}

// ------------------------------------------------------------------------------------ halo plan
struct Side {  // one side of the plan: per-neighbour ranks / counts / offsets + device index list
  std::vector<int> ranks;
  std::vector<int64_t> counts, offsets;
  int64_t total = 0;
  int64_t* idx_d = nullptr;  // concatenated index lists on the device
};

struct Halo {
  Comm* comm = nullptr;
  int eb = 8;  // element bytes
  int64_t nlocal = 0, nghost = 0;
  Side owners;  // my ghosts grouped by owning rank: indices into the ghost block
  Side ghosts;  // my owned dofs ghosted elsewhere, grouped by ghosting rank: local indices
  bool direct = false;       // ghosts numbered owner by owner: the ghost block is the owners-side message
  char* buf_owner = nullptr;  // owners-side message buffer (forward: recv, reverse: send), owners.total elements
  char* buf_ghost = nullptr;  // ghosts-side message buffer (forward: send, reverse: recv), ghosts.total elements
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  // LOCAL transport state
  int index = 0;                    // creation index within the rank (pairs halo objects across ranks)
  hipEvent_t ev_packed = nullptr;   // my message is complete in cur_send
  hipEvent_t ev_pulled = nullptr;   // my copies out of the peers' buffers have executed
  bool pulled_valid = false;
  const char* cur_send = nullptr;   // where my outgoing message lives for the exchange in flight
  int cur_dir = 0;                  // 0 forward, 1 reverse
  IpcState ipc;                     // PEER transport state
};

inline void side_free(Side& s) {
  if (s.idx_d) (void)hipFree(s.idx_d);
  s.idx_d = nullptr;
}

inline hipError_t side_init(Side& s, int nn, const int32_t* ranks, const int64_t* sizes, const int64_t* idx,
                            hipStream_t stream) {
  s.ranks.assign(ranks, ranks + nn);
  s.counts.assign(sizes, sizes + nn);
  s.offsets.assign(nn + 1, 0);
  for (int i = 0; i < nn; ++i) s.offsets[i + 1] = s.offsets[i] + s.counts[i];
  s.total = s.offsets[nn];
  if (s.total > 0) {
    hipError_t e = hipMalloc(&s.idx_d, s.total * sizeof(int64_t));
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(s.idx_d, idx, s.total * sizeof(int64_t), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <typename T>
inline hipError_t halo_launch(int mode, const void* in, void* out, const int64_t* index, int64_t count, int64_t offset,
                              hipStream_t s) {
  switch (mode) {
    case PACK: return launch_halo<T, PACK>((const T*)in, (T*)out, index, count, offset, s);
    case UNPACK_SET: return launch_halo<T, UNPACK_SET>((const T*)in, (T*)out, index, count, offset, s);
    default: return launch_halo<T, UNPACK_ADD>((const T*)in, (T*)out, index, count, offset, s);
  }
}

inline hipError_t halo_kernel_any(int eb, int mode, const void* in, void* out, const int64_t* index, int64_t count,
                                  int64_t offset, hipStream_t s) {
  return eb == 8 ? halo_launch<double>(mode, in, out, index, count, offset, s)
                 : halo_launch<float>(mode, in, out, index, count, offset, s);
}

// Post the receives and sends of one exchange (inside an open ncclGroup): send ``sside`` segments of
// ``sendbuf``, receive ``rside`` segments into ``recvbuf``.
inline ncclResult_t halo_post_rccl(Halo* h, const Side& sside, const char* sendbuf, const Side& rside, char* recvbuf) {
  Comm* c = h->comm;
  RcclApi& api = rccl();
  const ncclDataType_t dt = h->eb == 8 ? ncclFloat64 : ncclFloat32;
  ncclResult_t r = ncclSuccess;
  for (size_t i = 0; r == ncclSuccess && i < rside.ranks.size(); ++i)
    if (rside.counts[i] > 0)
      r = api.Recv(recvbuf + rside.offsets[i] * h->eb, (size_t)rside.counts[i], dt, rside.ranks[i], c->nccl, c->stream);
  for (size_t i = 0; r == ncclSuccess && i < sside.ranks.size(); ++i)
    if (sside.counts[i] > 0)
      r = api.Send(sendbuf + sside.offsets[i] * h->eb, (size_t)sside.counts[i], dt, sside.ranks[i], c->nccl, c->stream);
  return r;
}

// LOCAL transport, receiver side: pull every incoming segment out of the peer's current message.
inline hipError_t halo_pull_local(Halo* h, const Side& rside, char* recvbuf, int dir) {
  Comm* c = h->comm;
  for (size_t i = 0; i < rside.ranks.size(); ++i) {
    if (rside.counts[i] == 0) continue;
    const int peer_rank = rside.ranks[i];
    const auto& peers = c->world->halos[peer_rank];
    if (h->index >= (int)peers.size() || !peers[h->index]) return hipErrorInvalidValue;
    Halo* p = peers[h->index];
    if (!p->cur_send || p->cur_dir != dir) return hipErrorNotReady;  // the peer's *_begin has not been called
    // my segment inside the peer's outgoing message: the peer's send side lists me as a neighbour
    const Side& ps = (dir == 0) ? p->ghosts : p->owners;
    int64_t poff = -1;
    for (size_t k = 0; k < ps.ranks.size(); ++k)
      if (ps.ranks[k] == c->rank) {
        if (ps.counts[k] != rside.counts[i]) return hipErrorInvalidValue;
        poff = ps.offsets[k];
      }
    if (poff < 0) return hipErrorInvalidValue;
    hipError_t e = hipStreamWaitEvent(c->stream, p->ev_packed, 0);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(recvbuf + rside.offsets[i] * h->eb, p->cur_send + poff * h->eb, rside.counts[i] * h->eb,
                       hipMemcpyDeviceToDevice, c->stream);
    if (e != hipSuccess) return e;
  }
  hipError_t e = hipEventRecord(h->ev_pulled, c->stream);
  h->pulled_valid = true;
  return e;
}

// LOCAL transport, sender side: before overwriting my message buffer, wait until the peers that
// read the previous message out of it have done so.
inline hipError_t halo_wait_readers_local(Halo* h, const Side& sside) {
  Comm* c = h->comm;
  for (size_t i = 0; i < sside.ranks.size(); ++i) {
    const auto& peers = c->world->halos[sside.ranks[i]];
    if (h->index >= (int)peers.size() || !peers[h->index]) continue;
    Halo* p = peers[h->index];
    if (p->pulled_valid) {
      hipError_t e = hipStreamWaitEvent(c->stream, p->ev_pulled, 0);
      if (e != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

// ------------------------------------------------------------------------------------ PEER transport, host side
// A process that drives several ranks maps a neighbour's arena once, however many of its ranks border that neighbour
// (opening one HIP IPC handle twice in a process is not portable): process-wide table, reference counted.
struct IpcOpened {
  hipIpcMemHandle_t handle;
  void* ptr;
  int refs;
  std::vector<int> importers;  // devices of this process that may dereference ``ptr``
};
inline std::vector<IpcOpened>& ipc_opened_table() {  // guarded by local_worlds_mutex()
  static std::vector<IpcOpened> t;
  return t;
}
// ``importer``: the device whose kernels will use the mapping (the communicator's); ``exporter``: the device the arena lives on,
// as an ordinal of THIS process (-1: not visible here).  hipIpcMemLazyEnablePeerAccess enables peer access for the device that
// is current at the first open only: a second rank of this process on ANOTHER device that borders the same neighbour gets
// the cached pointer and must have peer access to the exporter enabled for itself, or its send / receive kernel faults
// (ADVICE r4).
inline hipError_t ipc_open_shared(const hipIpcMemHandle_t& handle, int importer, int exporter, void** out) {
  std::lock_guard<std::mutex> lock(local_worlds_mutex());
  for (IpcOpened& o : ipc_opened_table())
    if (!std::memcmp(&o.handle, &handle, sizeof handle)) {
      bool known = false;
      for (int d : o.importers) known = known || d == importer;
      if (!known) {
        if (exporter >= 0 && exporter != importer) {
          int cur = -1;
          hipError_t e = hipGetDevice(&cur);
          if (e == hipSuccess && cur != importer) e = hipSetDevice(importer);
          if (e == hipSuccess) {
            e = hipDeviceEnablePeerAccess(exporter, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) {
              (void)hipGetLastError();
              e = hipSuccess;
            }
          }
          if (cur >= 0 && cur != importer) (void)hipSetDevice(cur);
          if (e != hipSuccess) return e;
        }
        o.importers.push_back(importer);
      }
      ++o.refs;
      *out = o.ptr;
      return hipSuccess;
    }
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, handle, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return e;
  ipc_opened_table().push_back(IpcOpened{handle, p, 1, {importer}});
  *out = p;
  return hipSuccess;
}
inline void ipc_close_shared(void* p) {
  std::lock_guard<std::mutex> lock(local_worlds_mutex());
  auto& t = ipc_opened_table();
  for (size_t i = 0; i < t.size(); ++i)
    if (t[i].ptr == p) {
      if (--t[i].refs == 0) {
        (void)hipIpcCloseMemHandle(p);
        t.erase(t.begin() + (long)i);
      }
      return;
    }
}

inline void halo_ipc_free(Halo* h) {
  IpcState& st = h->ipc;
  for (void* p : st.opened) ipc_close_shared(p);
  st.opened.clear();
  ipc_role_free(st.send_fwd);
  ipc_role_free(st.recv_fwd);
  ipc_role_free(st.send_rev);
  ipc_role_free(st.recv_rev);
  if (st.arena) (void)hipFree(st.arena);
  if (st.status) (void)hipFree(st.status);
  if (st.ev_sent) (void)hipEventDestroy(st.ev_sent);
  if (st.join_counter) (void)hipFree(st.join_counter);
  st.join_counter = nullptr;
  st.arena = nullptr;
  st.status = nullptr;
  st.ev_sent = nullptr;
  st.connected = false;
}

// Arena layout: [flags: 4 kinds x nmax slots x 64 B][forward receive buffer: owners.total][reverse receive buffer: ghosts.total]
inline hipError_t halo_ipc_create(Halo* h) {
  IpcState& st = h->ipc;
  st.nmax = (int)std::max<size_t>(1, std::max(h->owners.ranks.size(), h->ghosts.ranks.size()));
  st.off_flags = 0;
  st.off_recv_fwd = ipc_align(4ll * st.nmax * kIpcFlagStride, 256);
  st.off_recv_rev = ipc_align(st.off_recv_fwd + h->owners.total * h->eb, 256);
  const int64_t bytes = ipc_align(st.off_recv_rev + h->ghosts.total * h->eb, 256) + 256;
  hipError_t e = ipc_arena_alloc(st, bytes);
  if (e == hipSuccess) e = hipMalloc(&st.status, ST_WORDS * sizeof(uint64_t));
  if (e == hipSuccess) e = hipMemset(st.status, 0, ST_WORDS * sizeof(uint64_t));
  if (e == hipSuccess) e = ipc_role_init(st.send_fwd, h->ghosts.counts, h->ghosts.offsets);
  if (e == hipSuccess) e = ipc_role_init(st.recv_rev, h->ghosts.counts, h->ghosts.offsets);
  if (e == hipSuccess) e = ipc_role_init(st.recv_fwd, h->owners.counts, h->owners.offsets);
  if (e == hipSuccess) e = ipc_role_init(st.send_rev, h->owners.counts, h->owners.offsets);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&st.ev_sent, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc(&st.join_counter, sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(st.join_counter, 0, sizeof(unsigned));
  int khz = 100000;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->comm->device);
  double seconds = 20.0;
  if (const char* v = std::getenv("FUS_IPC_SPIN_SECONDS")) seconds = std::atof(v) > 0 ? std::atof(v) : seconds;
  st.budget = (uint64_t)(seconds * 1e3 * khz);
  if (const char* v = std::getenv("FUS_IPC_FENCED")) st.fenced = std::atoi(v) != 0 ? 1 : 0;
  if (e == hipSuccess) e = hipDeviceSynchronize();  // arena zeroed before its handle can reach a peer
  return e;
}

inline int64_t halo_ipc_blob_bytes(const Halo* h) {
  return (int64_t)sizeof(IpcBlobHeader) + 3 * (int64_t)sizeof(int64_t) * (int64_t)(h->owners.ranks.size() + h->ghosts.ranks.size());
}

inline int halo_ipc_export(Halo* h, void* blob) {
  Comm* c = h->comm;
  IpcState& st = h->ipc;
  IpcBlobHeader hd;
  std::memset(&hd, 0, sizeof hd);
  hd.magic = kIpcMagic;
  hd.version = kIpcBlobVersion;
  hd.rank = c->rank;
  hd.elem_bytes = h->eb;
  hd.pid = (int64_t)getpid();
  hd.token = ipc_process_token();
  if (hipDeviceGetPCIBusId(hd.pci_bus_id, (int)sizeof hd.pci_bus_id, c->device) != hipSuccess) {
    (void)hipGetLastError();
    hd.pci_bus_id[0] = 0;
  }
  hd.pci_bus_id[sizeof hd.pci_bus_id - 1] = 0;
  hd.base = (uint64_t)(uintptr_t)st.arena;
  hd.arena_bytes = st.arena_bytes;
  hd.off_flags = st.off_flags;
  hd.off_recv_fwd = st.off_recv_fwd;
  hd.off_recv_rev = st.off_recv_rev;
  hd.n_owner = (int32_t)h->owners.ranks.size();
  hd.n_ghost = (int32_t)h->ghosts.ranks.size();
  hd.nmax = st.nmax;
  hd.device = c->device;
  const hipError_t e = hipIpcGetMemHandle(&hd.handle, st.arena);
  if (e != hipSuccess) {
    c->last_error = std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e);
    return -1;
  }
  char* out = static_cast<char*>(blob);
  std::memcpy(out, &hd, sizeof hd);
  int64_t* t = reinterpret_cast<int64_t*>(out + sizeof hd);
  for (const Side* s : {&h->owners, &h->ghosts})
    for (size_t k = 0; k < s->ranks.size(); ++k) {
      *t++ = s->ranks[k];
      *t++ = s->counts[k];
      *t++ = s->offsets[k];
    }
  return 0;
}

// ``blobs``: what fus_halo_ipc_export produced on the other ranks for THE SAME halo (same creation order), any order,
// at least one per neighbour rank (extra ones are ignored).  Maps the neighbours' arenas and fills the device tables.
inline int halo_ipc_connect(Halo* h, int nblobs, const void* const* blobs) {
  Comm* c = h->comm;
  IpcState& st = h->ipc;
  if (st.connected) {
    c->last_error = "halo already connected";
    return -1;
  }
  struct PeerView {
    const IpcBlobHeader* hd = nullptr;
    const int64_t* owners = nullptr;  // (rank, count, offset) triples
    const int64_t* ghosts = nullptr;
    char* arena = nullptr;
  };
  std::map<int, PeerView> views;
  for (int b = 0; b < nblobs; ++b) {
    if (!blobs[b]) continue;
    const auto* hd = static_cast<const IpcBlobHeader*>(blobs[b]);
    if (hd->magic != kIpcMagic || hd->version != kIpcBlobVersion || hd->elem_bytes != h->eb || hd->rank < 0 || hd->rank >= c->nranks) {
      c->last_error = "halo connect: malformed or mismatching blob";
      return -1;
    }
    PeerView v;
    v.hd = hd;
    v.owners = reinterpret_cast<const int64_t*>(static_cast<const char*>(blobs[b]) + sizeof(IpcBlobHeader));
    v.ghosts = v.owners + 3 * hd->n_owner;
    views[hd->rank] = v;
  }
  auto arena_of = [&](int rank, char** out) -> bool {
    auto it = views.find(rank);
    if (it == views.end()) {
      c->last_error = "halo connect: no blob from neighbour rank " + std::to_string(rank);
      return false;
    }
    PeerView& v = it->second;
    if (!v.arena) {
      const IpcProcessToken& me = ipc_process_token();
      if (v.hd->token.w[0] == me.w[0] && v.hd->token.w[1] == me.w[1]) {
        v.arena = reinterpret_cast<char*>((uintptr_t)v.hd->base);  // same address space (in-process ranks, self-neighbour)
        if (rank != c->rank) st.defer_recv = true;
      } else {
        // another process (normally another GPU): refuse up front what would otherwise fault inside a kernel.  The
        // exporter's device is identified by its PCI bus id -- ordinals are process-local -- and looked up among the
        // devices visible HERE; one that is not visible here cannot be checked and is left to hipIpcOpenMemHandle.
        int ndev = 0, can = 1, peer_dev = -1;
        if (v.hd->pci_bus_id[0] && hipGetDeviceCount(&ndev) == hipSuccess)
          for (int d = 0; d < ndev && peer_dev < 0; ++d) {
            char id[32] = {0};
            if (hipDeviceGetPCIBusId(id, (int)sizeof id, d) == hipSuccess && !strcasecmp(id, v.hd->pci_bus_id)) peer_dev = d;
          }
        if (peer_dev >= 0 && peer_dev != c->device && hipDeviceCanAccessPeer(&can, c->device, peer_dev) == hipSuccess && !can) {
          c->last_error = "device " + std::to_string(c->device) + " has no peer access to device " + std::string(v.hd->pci_bus_id) +
                          " (rank " + std::to_string(rank) + ")";
          return false;
        }
        (void)hipGetLastError();
        void* p = nullptr;
        const hipError_t e = ipc_open_shared(v.hd->handle, c->device, peer_dev, &p);
        if (e != hipSuccess) {
          c->last_error = "hipIpcOpenMemHandle (rank " + std::to_string(rank) + "): " + hipGetErrorString(e);
          return false;
        }
        st.opened.push_back(p);
        v.arena = static_cast<char*>(p);
      }
    }
    *out = v.arena;
    return true;
  };
  // slot of ``rank`` in a (rank, count, offset) triple list
  auto find_slot = [](const int64_t* triples, int n, int rank) {
    for (int k = 0; k < n; ++k)
      if (triples[3 * k] == rank) return k;
    return -1;
  };
  for (size_t i = 0; i < h->owners.ranks.size(); ++i) {  // my ghosts' owners: I receive forward, send reverse
    if (h->owners.counts[i] == 0) continue;
    char* pa = nullptr;
    if (!arena_of(h->owners.ranks[i], &pa)) return -1;
    const PeerView& v = views[h->owners.ranks[i]];
    const int js = find_slot(v.ghosts, v.hd->n_ghost, c->rank);
    if (js < 0 || v.ghosts[3 * js + 1] != h->owners.counts[i]) {
      c->last_error = "halo connect: rank " + std::to_string(h->owners.ranks[i]) + " does not list this rank with the same count";
      return -1;
    }
    IpcPeer& rf = st.recv_fwd.host_peers[i];
    rf.data = st.arena + st.off_recv_fwd;
    rf.flag_in = ipc_flag_ptr(st.arena, st.off_flags, st.nmax, ARRIVED_FWD, (int)i);
    rf.flag_out = ipc_flag_ptr(pa, v.hd->off_flags, v.hd->nmax, CREDIT_FWD, js);
    IpcPeer& sr = st.send_rev.host_peers[i];
    sr.data = pa + v.hd->off_recv_rev + v.ghosts[3 * js + 2] * h->eb;
    sr.flag_out = ipc_flag_ptr(pa, v.hd->off_flags, v.hd->nmax, ARRIVED_REV, js);
    sr.flag_in = ipc_flag_ptr(st.arena, st.off_flags, st.nmax, CREDIT_REV, (int)i);
  }
  for (size_t j = 0; j < h->ghosts.ranks.size(); ++j) {  // ranks ghosting my dofs: I send forward, receive reverse
    if (h->ghosts.counts[j] == 0) continue;
    char* pa = nullptr;
    if (!arena_of(h->ghosts.ranks[j], &pa)) return -1;
    const PeerView& v = views[h->ghosts.ranks[j]];
    const int is = find_slot(v.owners, v.hd->n_owner, c->rank);
    if (is < 0 || v.owners[3 * is + 1] != h->ghosts.counts[j]) {
      c->last_error = "halo connect: rank " + std::to_string(h->ghosts.ranks[j]) + " does not list this rank with the same count";
      return -1;
    }
    IpcPeer& sf = st.send_fwd.host_peers[j];
    sf.data = pa + v.hd->off_recv_fwd + v.owners[3 * is + 2] * h->eb;
    sf.flag_out = ipc_flag_ptr(pa, v.hd->off_flags, v.hd->nmax, ARRIVED_FWD, is);
    sf.flag_in = ipc_flag_ptr(st.arena, st.off_flags, st.nmax, CREDIT_FWD, (int)j);
    IpcPeer& rr = st.recv_rev.host_peers[j];
    rr.data = st.arena + st.off_recv_rev;
    rr.flag_in = ipc_flag_ptr(st.arena, st.off_flags, st.nmax, ARRIVED_REV, (int)j);
    rr.flag_out = ipc_flag_ptr(pa, v.hd->off_flags, v.hd->nmax, CREDIT_REV, is);
  }
  for (IpcRole* r : {&st.send_fwd, &st.recv_fwd, &st.send_rev, &st.recv_rev}) {
    const hipError_t e = ipc_role_upload(*r);
    if (e != hipSuccess) {
      c->last_error = hipGetErrorString(e);
      return -1;
    }
  }
  st.connected = true;
  return 0;
}

// ``on_stream``: the caller enqueues on the communicator's own stream (HaloApply's concurrent schedule): stream order is
// all the ordering there is, and no event is recorded (an event record is a marker with a cache write-back between the
// exchange kernels).
template <typename T>
inline hipError_t halo_ipc_post_recv(Halo* h, char* vecp, int dir, uint64_t seq, bool on_stream, bool inline_mode = false,
                                     hipStream_t run_on = nullptr) {
  Comm* c = h->comm;
  IpcState& st = h->ipc;
  hipStream_t rs = inline_mode ? run_on : c->stream2;  // inline: the caller's own stream (may be the null stream)
  T* vec = reinterpret_cast<T*>(vecp);
  const IpcRole& rr = dir == 0 ? st.recv_fwd : st.recv_rev;
  IpcJoin join{nullptr, 0, st.join_counter};
  if (c->join_armed && c->join_halo == h && c->join_dir == dir) {
    c->join_armed = false;
    c->join_halo = nullptr;
    if (rr.nchunks > 0 && c->stream2 == c->stream && !inline_mode) {  // this kernel is the last of the chain: it publishes the join flag
      join.flag = c->sync_words + 1;
      join.seq = ++c->sync_seq[1];
      c->join_inflight = join.seq;
    }
  }
  if (rr.nchunks > 0) {
    if (dir == 1)
      hipLaunchKernelGGL((ipc_recv_kernel<T, UNPACK_ADD, true>), dim3(rr.nchunks), dim3(ipc_threads()), 0, rs, vec, h->ghosts.idx_d,
                         (int64_t)0, rr.chunks, rr.peers, rr.counters, st.status, seq, st.budget, join, st.fenced);
    else if (h->direct)
      hipLaunchKernelGGL((ipc_recv_kernel<T, UNPACK_SET, false>), dim3(rr.nchunks), dim3(ipc_threads()), 0, rs, vec, h->owners.idx_d,
                         h->nlocal, rr.chunks, rr.peers, rr.counters, st.status, seq, st.budget, join, st.fenced);
    else
      hipLaunchKernelGGL((ipc_recv_kernel<T, UNPACK_SET, true>), dim3(rr.nchunks), dim3(ipc_threads()), 0, rs, vec, h->owners.idx_d,
                         h->nlocal, rr.chunks, rr.peers, rr.counters, st.status, seq, st.budget, join, st.fenced);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  st.done_recorded = !on_stream;
  return on_stream ? hipSuccess : hipEventRecord(h->ev_done, rs);
}

template <typename T>
inline hipError_t halo_ipc_post(Halo* h, char* vecp, int dir, bool on_stream, bool inline_mode = false, hipStream_t run_on = nullptr) {
  Comm* c = h->comm;
  IpcState& st = h->ipc;
  hipStream_t ss = inline_mode ? run_on : c->stream;
  T* vec = reinterpret_cast<T*>(vecp);
  const uint64_t seq = ++st.seq[dir];
  const IpcRole& sr = dir == 0 ? st.send_fwd : st.send_rev;
  // a lazily posted fork: the first send kernel on the communicator's stream waits for the fork flag itself; where there
  // is none to carry the wait (no neighbours on this side), a wait kernel does
  IpcGate gate{nullptr, 0, nullptr};
  if (c->gate_pending && !inline_mode) {
    if (sr.nchunks > 0) {
      gate = IpcGate{c->sync_words + 0, c->gate_pending, c->sync_words + 2};
      c->gate_pending = 0;
    } else {
      const hipError_t e = comm_flush_gate(c);
      if (e != hipSuccess) return e;
    }
  }
  if (sr.nchunks > 0) {
    if (dir == 0)  // owned entries listed in ghosts.idx -> the ghosting ranks
      hipLaunchKernelGGL((ipc_send_kernel<T, true>), dim3(sr.nchunks), dim3(ipc_threads()), 0, ss, vec, h->ghosts.idx_d, (int64_t)0,
                         sr.chunks, sr.peers, sr.counters, st.status, seq, st.budget, gate, st.fenced);
    else if (h->direct)  // ghost block, already grouped by owner -> the owners
      hipLaunchKernelGGL((ipc_send_kernel<T, false>), dim3(sr.nchunks), dim3(ipc_threads()), 0, ss, vec, h->owners.idx_d, h->nlocal,
                         sr.chunks, sr.peers, sr.counters, st.status, seq, st.budget, gate, st.fenced);
    else
      hipLaunchKernelGGL((ipc_send_kernel<T, true>), dim3(sr.nchunks), dim3(ipc_threads()), 0, ss, vec, h->owners.idx_d, h->nlocal,
                         sr.chunks, sr.peers, sr.counters, st.status, seq, st.budget, gate, st.fenced);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  st.sent_recorded = !on_stream;
  if (!on_stream) {
    hipError_t e = hipEventRecord(st.ev_sent, ss);
    if (e != hipSuccess) return e;
  }
  if (st.defer_recv) {  // ranks of ONE process: the receive kernel is posted by *_end (see IpcState::defer_recv)
    st.pending[dir] = seq;
    return hipSuccess;
  }
  return halo_ipc_post_recv<T>(h, vecp, dir, seq, on_stream, inline_mode, run_on);
}

// failures seen by this halo's kernels (0 = healthy); synchronises the communicator's streams
//   out8 = {failed waits (time-outs + poisoned flags read), forward exchanges posted, reverse exchanges posted, arena
//           memory kind, time-outs, poisoned flags read (a neighbour's halo had failed), dead (0 / 1), 0}
inline int halo_ipc_status(Halo* h, int64_t* out8) {
  Comm* c = h->comm;
  uint64_t w[ST_WORDS] = {0};
  // the whole device: a stand-alone exchange runs its kernels on the CALLER's stream (halo_exchange_inline), not on the
  // communicator's
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(w, h->ipc.status, sizeof w, hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    c->last_error = hipGetErrorString(e);
    return -1;
  }
  out8[0] = (int64_t)(w[ST_TIMEOUTS] + w[ST_POISONED]);
  out8[1] = (int64_t)h->ipc.seq[0];
  out8[2] = (int64_t)h->ipc.seq[1];
  out8[3] = h->ipc.memory_kind;
  out8[4] = (int64_t)w[ST_TIMEOUTS];
  out8[5] = (int64_t)w[ST_POISONED];
  out8[6] = (int64_t)w[ST_DEAD];
  out8[7] = h->ipc.fenced;
  return 0;
}

// Failed device-side waits of EVERY halo object of the communicator plus of its fork / join kernels (0 = healthy): what a
// time loop checks before it trusts its result.  Synchronises the communicator's streams (one small copy per halo).
inline int comm_health(Comm* c, int64_t* failures, int64_t* detail3 = nullptr) {
  *failures = 0;
  int64_t d[3] = {0, 0, 0};  // time-outs of halo waits, poisoned flags read, time-outs of fork / join waits
  hipError_t e = hipDeviceSynchronize();  // exchange kernels may have run on the communicator's stream(s) or on a caller's
  if (e == hipSuccess && c->sync_words) {
    uint64_t w = 0;
    e = hipMemcpy(&w, c->sync_words + 2 + ST_TIMEOUTS, sizeof w, hipMemcpyDeviceToHost);
    *failures += (int64_t)w;
    d[2] = (int64_t)w;
  }
  if (e == hipSuccess && c->kind == Comm::PEER)
    for (Halo* h : c->halos) {
      uint64_t w[ST_WORDS] = {0};
      if (!h->ipc.status) continue;
      e = hipMemcpy(w, h->ipc.status, sizeof w, hipMemcpyDeviceToHost);
      if (e != hipSuccess) break;
      *failures += (int64_t)(w[ST_TIMEOUTS] + w[ST_POISONED]);
      d[0] += (int64_t)w[ST_TIMEOUTS];
      d[1] += (int64_t)w[ST_POISONED];
    }
  if (e != hipSuccess) {
    c->last_error = hipGetErrorString(e);
    return -1;
  }
  if (detail3) std::memcpy(detail3, d, sizeof d);
  return 0;
}

#define FUS_H(e_)                                \
  do {                                           \
    hipError_t _e = (e_);                        \
    if (_e != hipSuccess) {                      \
      c->last_error = hipGetErrorString(_e);     \
      return -1;                                 \
    }                                            \
  } while (0)

// dir 0: forward (owners -> ghosts, overwrite)   cuda/scatterer.py:191-277
// dir 1: reverse (ghosts -> owners, add)         cuda/scatterer.py:104-188
//
// Begin ``nh`` exchanges (one per vector, halos of ONE communicator) as one unit: one event edge from the
// caller's stream, the packs, ONE ncclGroup with every receive and send of every vector (two messages to
// the same peer are matched in issue order, which is the same on both sides), the unpacks, each halo's
// "done" event.  The RK4 stage forward-scatters two vectors (u_n, v_n): one RCCL launch instead of two.
inline int halo_begin_group(Halo* const* hs, void* const* buffers, int nh, hipStream_t stream, int dir) {
  if (nh <= 0) return 0;
  Comm* c = hs[0]->comm;
  bool any = false;
  for (int k = 0; k < nh; ++k) {
    if (hs[k]->comm != c) {
      c->last_error = "halo group: the halos belong to different communicators";
      return -1;
    }
    any = any || hs[k]->owners.total > 0 || hs[k]->ghosts.total > 0;
  }
  if (!any) return 0;  // no neighbours: nothing to order, nothing to move
  if (nh > 8) {
    c->last_error = "halo group: at most 8 vectors";
    return -1;
  }
  if (c->kind == Comm::PEER)
    for (int k = 0; k < nh; ++k)
      if (!hs[k]->ipc.connected) {
        c->last_error = "halo not connected: exchange the blobs of fus_halo_ipc_export and call fus_halo_ipc_connect first";
        return -1;
      }
  if (c->kind == Comm::PEER) {
    // the caller may BE on the communicator's stream (HaloApply's concurrent schedule): then stream order is all there is
    const bool on_stream = stream == c->stream && stream == c->stream2;
    if (!on_stream) {
      FUS_H(comm_flush_gate(c));
      FUS_H(hipEventRecord(hs[0]->ev_ready, stream));
      if (stream != c->stream) FUS_H(hipStreamWaitEvent(c->stream, hs[0]->ev_ready, 0));
      if (stream != c->stream2 && c->stream2 != c->stream) FUS_H(hipStreamWaitEvent(c->stream2, hs[0]->ev_ready, 0));
    }
    if (c->join_armed && !c->join_halo) {  // fus_comm_arm_join: the last receive kernel of THIS call carries the join
      c->join_halo = hs[nh - 1];
      c->join_dir = dir;
    }
    for (int k = 0; k < nh; ++k) {
      Halo* h = hs[k];
      FUS_H(h->eb == 8 ? halo_ipc_post<double>(h, static_cast<char*>(buffers[k]), dir, on_stream)
                       : halo_ipc_post<float>(h, static_cast<char*>(buffers[k]), dir, on_stream));
    }
    return 0;
  }
  FUS_H(hipEventRecord(hs[0]->ev_ready, stream));
  FUS_H(hipStreamWaitEvent(c->stream, hs[0]->ev_ready, 0));
  const char* sendbuf[8];
  char* recvbuf[8];
  // ---- pack
  for (int k = 0; k < nh; ++k) {
    Halo* h = hs[k];
    char* vec = static_cast<char*>(buffers[k]);
    char* ghost_block = vec + h->nlocal * h->eb;
    const Side& sside = dir == 0 ? h->ghosts : h->owners;
    if (c->kind == Comm::LOCAL) FUS_H(halo_wait_readers_local(h, sside));
    if (dir == 0) {
      FUS_H(halo_kernel_any(h->eb, PACK, vec, h->buf_ghost, h->ghosts.idx_d, h->ghosts.total, 0, c->stream));
      sendbuf[k] = h->buf_ghost;
    } else if (h->direct) {
      sendbuf[k] = ghost_block;
    } else {
      FUS_H(halo_kernel_any(h->eb, PACK, vec, h->buf_owner, h->owners.idx_d, h->owners.total, h->nlocal, c->stream));
      sendbuf[k] = h->buf_owner;
    }
    recvbuf[k] = dir == 0 ? (h->direct ? ghost_block : h->buf_owner) : h->buf_ghost;
  }
  if (c->kind == Comm::RCCL) {
    RcclApi& api = rccl();
    ncclResult_t r = api.GroupStart();
    for (int k = 0; r == ncclSuccess && k < nh; ++k) {
      Halo* h = hs[k];
      r = halo_post_rccl(h, dir == 0 ? h->ghosts : h->owners, sendbuf[k], dir == 0 ? h->owners : h->ghosts, recvbuf[k]);
    }
    const ncclResult_t r2 = api.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) {
      c->last_error = std::string("RCCL: ") + api.GetErrorString(r);
      return -1;
    }
    // ---- unpack
    for (int k = 0; k < nh; ++k) {
      Halo* h = hs[k];
      char* vec = static_cast<char*>(buffers[k]);
      if (dir == 0) {
        if (!h->direct)
          FUS_H(halo_kernel_any(h->eb, UNPACK_SET, h->buf_owner, vec, h->owners.idx_d, h->owners.total, h->nlocal, c->stream));
      } else {
        FUS_H(halo_kernel_any(h->eb, UNPACK_ADD, h->buf_ghost, vec, h->ghosts.idx_d, h->ghosts.total, 0, c->stream));
      }
      FUS_H(hipEventRecord(h->ev_done, c->stream));
    }
  } else {
    for (int k = 0; k < nh; ++k) {
      hs[k]->cur_send = sendbuf[k];
      hs[k]->cur_dir = dir;
      FUS_H(hipEventRecord(hs[k]->ev_packed, c->stream));
    }
  }
  return 0;
}

// begin + end of ONE exchange in one call (scatter_forward(buffer) / scatter_reverse(buffer) of the reference's closures called
// stand-alone: set-up exchanges, u_sol(with_ghosts), the non-overlapped stage).  PEER: both kernels run on the CALLER's stream,
// in stream order with what precedes and follows them -- no event edge to the communicator's stream and back (a stand-alone
// exchange has nothing to overlap with; profiles/r04l_scatter_alone.log).  A receive that was deferred (ranks of one process)
// is posted here too, so a host driving several ranks calls begin / end separately instead.
inline int halo_exchange_inline(Halo* h, void* buffer, hipStream_t stream, int dir) {
  Comm* c = h->comm;
  if (h->owners.total == 0 && h->ghosts.total == 0) return 0;
  if (c->kind != Comm::PEER) return 1;  // not handled here
  if (!h->ipc.connected) {
    c->last_error = "halo not connected: exchange the blobs of fus_halo_ipc_export and call fus_halo_ipc_connect first";
    return -1;
  }
  char* vec = static_cast<char*>(buffer);
  FUS_H(h->eb == 8 ? halo_ipc_post<double>(h, vec, dir, true, true, stream) : halo_ipc_post<float>(h, vec, dir, true, true, stream));
  if (h->ipc.defer_recv && h->ipc.pending[dir]) {
    const uint64_t seq = h->ipc.pending[dir];
    h->ipc.pending[dir] = 0;
    FUS_H(h->eb == 8 ? halo_ipc_post_recv<double>(h, vec, dir, seq, true, true, stream) : halo_ipc_post_recv<float>(h, vec, dir, seq, true, true, stream));
  }
  return 0;
}

inline int halo_begin(Halo* h, void* buffer, hipStream_t stream, int dir) {
  return halo_begin_group(&h, &buffer, 1, stream, dir);
}

inline int halo_end(Halo* h, void* buffer, hipStream_t stream, int dir) {
  Comm* c = h->comm;
  if (h->owners.total == 0 && h->ghosts.total == 0) return 0;
  char* vec = static_cast<char*>(buffer);
  if (c->kind == Comm::LOCAL) {
    char* ghost_block = vec + h->nlocal * h->eb;
    const Side& rside = dir == 0 ? h->owners : h->ghosts;
    char* recvbuf = dir == 0 ? (h->direct ? ghost_block : h->buf_owner) : h->buf_ghost;
    FUS_H(halo_pull_local(h, rside, recvbuf, dir));
    if (dir == 0) {
      if (!h->direct)
        FUS_H(halo_kernel_any(h->eb, UNPACK_SET, h->buf_owner, vec, h->owners.idx_d, h->owners.total, h->nlocal, c->stream));
    } else {
      FUS_H(halo_kernel_any(h->eb, UNPACK_ADD, h->buf_ghost, vec, h->ghosts.idx_d, h->ghosts.total, 0, c->stream));
    }
    FUS_H(hipEventRecord(h->ev_done, c->stream));
  }
  if (c->kind == Comm::PEER) {
    const bool on_stream = stream == c->stream && stream == c->stream2;
    if (h->ipc.defer_recv && h->ipc.pending[dir]) {
      const uint64_t seq = h->ipc.pending[dir];
      h->ipc.pending[dir] = 0;
      FUS_H(h->eb == 8 ? halo_ipc_post_recv<double>(h, vec, dir, seq, on_stream) : halo_ipc_post_recv<float>(h, vec, dir, seq, on_stream));
    }
    if (on_stream) return 0;  // the caller is on the exchange kernels' stream: already ordered
    // begin was called on the communicator's stream, end from elsewhere: the events were not recorded then -- now is as good
    if (!h->ipc.sent_recorded) FUS_H(hipEventRecord(h->ipc.ev_sent, c->stream));
    if (!h->ipc.done_recorded) FUS_H(hipEventRecord(h->ev_done, c->stream2));
    h->ipc.sent_recorded = h->ipc.done_recorded = true;
    if (stream != c->stream) FUS_H(hipStreamWaitEvent(stream, h->ipc.ev_sent, 0));  // the send kernel has read the vector
    if (stream == c->stream2) return 0;
  }
  FUS_H(hipStreamWaitEvent(stream, h->ev_done, 0));
#undef FUS_H
  return 0;
}

}  // namespace fus
