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
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>

#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "halo.hpp"

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
  enum Kind { RCCL = 0, LOCAL = 1 } kind = RCCL;
  int rank = 0, nranks = 1, device = 0;
  ncclComm_t nccl = nullptr;
  std::shared_ptr<LocalWorld> world;
  hipStream_t stream = nullptr;  // high priority: small exchange kernels between big operator kernels
  std::string last_error;
};

inline std::map<int, std::weak_ptr<LocalWorld>>& local_worlds() {
  static std::map<int, std::weak_ptr<LocalWorld>> m;
  return m;
}

inline hipError_t comm_make_stream(Comm* c) {
  int lo = 0, hi = 0;
  hipError_t e = hipGetDevice(&c->device);
  if (e != hipSuccess) return e;
  e = hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically lowest = highest priority
  if (e != hipSuccess) return e;
  return hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi);
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
  FUS_H(hipEventRecord(hs[0]->ev_ready, stream));
  FUS_H(hipStreamWaitEvent(c->stream, hs[0]->ev_ready, 0));
  const char* sendbuf[8];
  char* recvbuf[8];
  if (nh > 8) {
    c->last_error = "halo group: at most 8 vectors";
    return -1;
  }
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
  FUS_H(hipStreamWaitEvent(stream, h->ev_done, 0));
#undef FUS_H
  return 0;
}

}  // namespace fus
