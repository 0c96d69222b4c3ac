// Probe of the HIP primitives the IPC halo transport rests on, between TWO real processes sharing one GPU
// (the pool gives one GPU per box; on an 8-GPU node the same calls cross xGMI).
//
//   hipcc --offload-arch=gfx950 -O2 -o ipc_probe tools/ipc_probe.hip && ./ipc_probe
//
// What it answers (each line printed by the parent):
//   1. can fine-grained / uncached device memory be exported with hipIpcGetMemHandle and opened by a peer
//   2. kernel push into the peer's mapped buffer + __threadfence_system + sequence flag, peer spins (bounded) on its
//      local flag: correct?  round-trip latency?
//   3. the same with hipMemcpyAsync into the mapped buffer as the data mover
//   4. interprocess events (hipEventInterprocess): round-trip latency
//   5. hipStreamWriteValue64 / hipStreamWaitValue64 on the mapped memory: supported? latency?
// Every device-side wait is bounded by a wall-clock budget, so a broken primitive fails instead of hanging the box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>

#define CK(x)                                                                                     \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) {                                                                       \
      fprintf(stderr, "[%d] %s:%d %s -> %s\n", g_rank, __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(3);                                                                                    \
    }                                                                                             \
  } while (0)

static int g_rank = 0;
static int g_sock = -1;

static void xsend(const void* p, size_t n) {
  if (write(g_sock, p, n) != (ssize_t)n) { perror("write"); exit(4); }
}
static void xrecv(void* p, size_t n) {
  size_t got = 0;
  while (got < n) {
    ssize_t r = read(g_sock, (char*)p + got, n - got);
    if (r <= 0) { perror("read"); exit(4); }
    got += r;
  }
}
static void hbarrier() {
  char c = 1;
  xsend(&c, 1);
  xrecv(&c, 1);
}

struct Arena {          // one per process, mapped by the peer
  uint64_t flag[8];     // [0] data-arrived sequence, [1] credit, ...
  uint64_t status[8];   // [0] timeouts seen by my spin kernels
  double data[1 << 17]; // 1 MiB message
};

__device__ inline bool spin_until(volatile uint64_t* f, uint64_t want, uint64_t budget_ticks) {
  const uint64_t t0 = wall_clock64();
  while (__hip_atomic_load((uint64_t*)f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want)
    if (wall_clock64() - t0 > budget_ticks) return false;
  return true;
}

// sender: write n doubles (value = base + i) into the peer's data, then publish seq in the peer's flag[slot]
__global__ void push_kernel(double* peer_data, uint64_t* peer_flag, unsigned* counter, int n, double base, uint64_t seq) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) peer_data[i] = base + i;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = atomicAdd(counter, 1u);
    if (done == gridDim.x - 1) {
      *counter = 0;
      __hip_atomic_store(peer_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void flag_kernel(uint64_t* peer_flag, uint64_t seq) {
  __threadfence_system();
  __hip_atomic_store(peer_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// receiver: wait for seq in my flag, then check the data; errors and time-outs counted in status
__global__ void wait_check_kernel(const double* data, uint64_t* flag, uint64_t* status, int n, double base, uint64_t seq,
                                  uint64_t budget) {
  __shared__ int ok;
  if (threadIdx.x == 0) ok = spin_until(flag, seq, budget) ? 1 : 0;
  __syncthreads();
  if (!ok) {
    if (threadIdx.x == 0) atomicAdd((unsigned long long*)&status[0], 1ull);
    return;
  }
  unsigned long long bad = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double v = __builtin_nontemporal_load(&data[i]);
    if (v != base + i) ++bad;
  }
  if (bad) atomicAdd((unsigned long long*)&status[1], bad);
}

__global__ void wait_kernel(uint64_t* flag, uint64_t* status, uint64_t seq, uint64_t budget) {
  if (!spin_until(flag, seq, budget)) atomicAdd((unsigned long long*)&status[0], 1ull);
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  int sv[2];
  if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { perror("socketpair"); return 2; }
  pid_t pid = fork();  // before any HIP call
  g_rank = pid == 0 ? 1 : 0;
  g_sock = sv[g_rank];
  close(sv[1 - g_rank]);
  alarm(240);  // host-side bound of the whole probe

  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int rate_khz = 0;
  CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  const uint64_t budget = (uint64_t)rate_khz * 1000ull * 5ull;  // 5 s of wall_clock64 ticks
  if (g_rank == 0) printf("wall clock rate %d kHz; spin budget 5 s\n", rate_khz);

  for (int kind = 0; kind < 3; ++kind) {
    const char* kname = kind == 0 ? "hipDeviceMallocFinegrained" : kind == 1 ? "hipDeviceMallocUncached" : "hipMalloc (coarse)";
    Arena* mine = nullptr;
    hipError_t e = kind == 0   ? hipExtMallocWithFlags((void**)&mine, sizeof(Arena), hipDeviceMallocFinegrained)
                   : kind == 1 ? hipExtMallocWithFlags((void**)&mine, sizeof(Arena), hipDeviceMallocUncached)
                               : hipMalloc((void**)&mine, sizeof(Arena));
    int ok = e == hipSuccess;
    hipIpcMemHandle_t hm, hp;
    memset(&hm, 0, sizeof hm);
    if (ok) {
      CK(hipMemset(mine, 0, sizeof(Arena)));
      e = hipIpcGetMemHandle(&hm, mine);
      ok = e == hipSuccess;
    }
    if (!ok) fprintf(stderr, "[%d] %s: alloc/export failed: %s\n", g_rank, kname, hipGetErrorString(e));
    int peer_ok = 0;
    xsend(&ok, sizeof ok);
    xrecv(&peer_ok, sizeof peer_ok);
    xsend(&hm, sizeof hm);
    xrecv(&hp, sizeof hp);
    if (!(ok && peer_ok)) {
      if (g_rank == 0) printf("1. %-28s export: FAILED\n", kname);
      continue;
    }
    Arena* peer = nullptr;
    e = hipIpcOpenMemHandle((void**)&peer, hp, hipIpcMemLazyEnablePeerAccess);
    ok = e == hipSuccess;
    xsend(&ok, sizeof ok);
    xrecv(&peer_ok, sizeof peer_ok);
    if (!(ok && peer_ok)) {
      if (g_rank == 0) printf("1. %-28s export ok, open: FAILED (%s)\n", kname, hipGetErrorString(e));
      continue;
    }
    if (g_rank == 0) printf("1. %-28s export + open: ok\n", kname);
    unsigned* counter;
    CK(hipMalloc((void**)&counter, 4));
    CK(hipMemset(counter, 0, 4));
    double* src;
    const int n = 1 << 17;
    CK(hipMalloc((void**)&src, n * 8));
    uint64_t* dummy;
    CK(hipMalloc((void**)&dummy, 8));
    CK(hipDeviceSynchronize());
    hbarrier();

    // ---- 2. kernel push ping-pong: rank 0 pushes seq 2k+1, rank 1 waits+checks then pushes 2k+2, rank 0 waits+checks
    for (int mover = 0; mover < 2; ++mover) {
      CK(hipMemset(mine->flag, 0, sizeof mine->flag));
      CK(hipDeviceSynchronize());
      hbarrier();
      const double t0 = now_us();
      for (int k = 0; k < reps; ++k) {
        const uint64_t a = 2 * k + 1, b = 2 * k + 2;
        const uint64_t mine_seq = g_rank == 0 ? a : b, theirs = g_rank == 0 ? b : a;
        auto send = [&](uint64_t seq) {
          if (mover == 0) {
            push_kernel<<<64, 256, 0, s>>>(peer->data, &peer->flag[0], counter, n, (double)seq, seq);
          } else {
            // data mover = the runtime's copy (SDMA between devices; a blit kernel inside one device)
            push_kernel<<<64, 256, 0, s>>>(src, dummy, counter, n, (double)seq, 0);
            CK(hipMemcpyAsync(peer->data, src, n * 8, hipMemcpyDeviceToDevice, s));
            flag_kernel<<<1, 1, 0, s>>>(&peer->flag[0], seq);
          }
        };
        if (g_rank == 0) {
          send(mine_seq);
          wait_check_kernel<<<64, 256, 0, s>>>(mine->data, &mine->flag[0], mine->status, n, (double)theirs, theirs, budget);
        } else {
          wait_check_kernel<<<64, 256, 0, s>>>(mine->data, &mine->flag[0], mine->status, n, (double)theirs, theirs, budget);
          send(mine_seq);
        }
      }
      CK(hipStreamSynchronize(s));
      const double t1 = now_us();
      uint64_t st[2];
      CK(hipMemcpy(st, mine->status, sizeof st, hipMemcpyDeviceToHost));
      uint64_t pst[2];
      xsend(st, sizeof st);
      xrecv(pst, sizeof pst);
      if (g_rank == 0)
        printf("%d. %-28s %s: %d round trips of 1 MiB each way, %.1f us per round trip; time-outs %llu/%llu, wrong values %llu/%llu%s\n",
               2 + mover, kname, mover == 0 ? "kernel push + flag" : "memcpyAsync + flag kernel", reps, (t1 - t0) / reps,
               (unsigned long long)st[0], (unsigned long long)pst[0], (unsigned long long)st[1], (unsigned long long)pst[1],
               (st[0] | pst[0] | st[1] | pst[1]) ? "  <-- BROKEN" : "");
      CK(hipMemset(mine->status, 0, sizeof mine->status));
      CK(hipDeviceSynchronize());
      hbarrier();
    }

    // ---- 5. stream memory operations on the mapped flags
    {
      int can = 0;
      (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
      CK(hipMemset(mine->flag, 0, sizeof mine->flag));
      CK(hipDeviceSynchronize());
      hbarrier();
      hipError_t e1 = hipSuccess;
      const double t0 = now_us();
      const int r5 = reps < 50 ? reps : 50;
      for (int k = 0; k < r5 && e1 == hipSuccess; ++k) {
        const uint64_t a = 2 * k + 1, b = 2 * k + 2;
        if (g_rank == 0) {
          e1 = hipStreamWriteValue64(s, &peer->flag[1], a, 0);
          if (e1 == hipSuccess) e1 = hipStreamWaitValue64(s, &mine->flag[1], b, hipStreamWaitValueGte, ~0ull);
        } else {
          e1 = hipStreamWaitValue64(s, &mine->flag[1], a, hipStreamWaitValueGte, ~0ull);
          if (e1 == hipSuccess) e1 = hipStreamWriteValue64(s, &peer->flag[1], b, 0);
        }
      }
      int fine = e1 == hipSuccess, pfine = 0;
      xsend(&fine, sizeof fine);
      xrecv(&pfine, sizeof pfine);
      if (fine && pfine) {
        // bounded host wait: if the stream ops never complete, give up after 20 s instead of hanging
        const double lim = now_us() + 20e6;
        hipError_t q;
        while ((q = hipStreamQuery(s)) == hipErrorNotReady && now_us() < lim) usleep(100);
        const double t1 = now_us();
        if (g_rank == 0)
          printf("5. %-28s hipStreamWrite/WaitValue64 (attribute CanUseStreamWaitValue=%d): %s, %.1f us per round trip\n", kname, can,
                 q == hipSuccess ? "ok" : "DID NOT COMPLETE", (t1 - t0) / r5);
        if (q != hipSuccess) { fprintf(stderr, "[%d] stream value ops stuck; leaving\n", g_rank); _exit(5); }
      } else if (g_rank == 0) {
        printf("5. %-28s hipStreamWrite/WaitValue64: call failed (%s)\n", kname, hipGetErrorString(e1));
      }
      hbarrier();
    }
    CK(hipIpcCloseMemHandle(peer));
    hbarrier();
    CK(hipFree(mine));
    CK(hipFree(counter));
    CK(hipFree(src));
    CK(hipFree(dummy));
  }

  // ---- 4. interprocess events
  {
    hipEvent_t mine_ev, peer_ev;
    hipError_t e = hipEventCreateWithFlags(&mine_ev, hipEventDisableTiming | hipEventInterprocess);
    hipIpcEventHandle_t hm, hp;
    memset(&hm, 0, sizeof hm);
    if (e == hipSuccess) e = hipIpcGetEventHandle(&hm, mine_ev);
    int ok = e == hipSuccess, pok = 0;
    xsend(&ok, sizeof ok);
    xrecv(&pok, sizeof pok);
    xsend(&hm, sizeof hm);
    xrecv(&hp, sizeof hp);
    if (ok && pok) {
      e = hipIpcOpenEventHandle(&peer_ev, hp);
      ok = e == hipSuccess;
    }
    xsend(&ok, sizeof ok);
    xrecv(&pok, sizeof pok);
    if (!(ok && pok)) {
      if (g_rank == 0) printf("4. interprocess events: FAILED (%s)\n", hipGetErrorString(e));
    } else {
      // host hand-shake per hop (a wait captures the most recent record CALL): rank 0 records, tells rank 1 over the
      // socket, rank 1 waits on it in-stream, records its own, tells rank 0, ...
      const int r4 = reps < 100 ? reps : 100;
      hbarrier();
      const double t0 = now_us();
      char c = 0;
      for (int k = 0; k < r4; ++k) {
        if (g_rank == 0) {
          CK(hipEventRecord(mine_ev, s));
          xsend(&c, 1);
          xrecv(&c, 1);
          CK(hipStreamWaitEvent(s, peer_ev, 0));
        } else {
          xrecv(&c, 1);
          CK(hipStreamWaitEvent(s, peer_ev, 0));
          CK(hipEventRecord(mine_ev, s));
          xsend(&c, 1);
        }
      }
      CK(hipStreamSynchronize(s));
      const double t1 = now_us();
      if (g_rank == 0) printf("4. interprocess events: ok, %.1f us per round trip (host hand-shake over a socket included)\n", (t1 - t0) / r4);
    }
    hbarrier();
  }
  if (g_rank == 0) {
    int st = 0;
    waitpid(pid, &st, 0);
    printf("child exit status %d\n", WIFEXITED(st) ? WEXITSTATUS(st) : -1);
  }
  return 0;
}
