// Mass apply without atomics: the TRANSPOSED dofmap ("which (entity, local index) entries touch dof d"), one thread per
// touched dof.  Same sum as numba-cpu/operators.py:19-68 / cuda/operators.py:18-70,
//     y[d] += sum over the entries e = (entity c, local i) with dofmap[c, i] == d of  x[d] * detJ[c, i] * constants[c],
// and the same bytes (detJ once, one 4-byte index per entry, x once, y read-modify-write once), but every dof is
// finished by ONE thread: plain coalesced loads / stores of x and y instead of a scattered float atomic per shared
// dof -- the request rate of those atomics is what bounds mass_plan_kernel at 0.44-0.45 of the HBM roofline (DESIGN
// 3.4).  The entries of a dof are visited in ascending (entity, local index) order, the order of the reference's serial
// loop: the result does not depend on scheduling (bitwise reproducible run to run, unlike the atomic kernels).
//
// Plan (built once per dofmap on the device, fus_mass_gather_plan_build; lives in a caller-owned workspace):
//   header | rows[nrows] (touched dofs, ascending; omitted when they are 0..nrows-1) | len[nrows] (uint8: entries of the
//   row) | block_base[nblocks + 1] (first entry of each 256-row block) | entries[nent * N] (entry ids c * N + i, sorted
//   by dof, ties ascending)
// The row pointer of a thread is block_base[block] + the exclusive prefix sum of len[] inside the block (wave scan +
// one LDS hand-off): 1 byte per dof instead of a 4-byte row pointer.
#pragma once

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <type_traits>

#include "vecops.hpp"

namespace fus {

constexpr int kGatherThreads = 256;
constexpr int64_t kGatherMagic = 0x4655534d47415431LL;  // "FUSMGAT1"
constexpr int kGatherHeaderBytes = 256;

struct GatherHeader {
  int64_t magic, nent, N, nrows, dense, nblocks, bytes, max_len;
  int64_t off_rows, off_len, off_base, off_entries;
};

struct GatherView {
  int64_t nrows, nblocks;
  const int32_t* rows;  // nullptr: dense (row r is dof r)
  const uint8_t* len;
  const int32_t* base;
  const int32_t* entries;
};

inline int64_t gather_align(int64_t v) { return (v + 255) / 256 * 256; }

// workspace layout: sizes depend on (nent, N) and on the length of the dof vector (rows <= min(entries, ndofs))
inline void gather_layout(int64_t nent, int N, int64_t ndofs, GatherHeader* h) {
  const int64_t total = nent * N;
  const int64_t maxrows = total < ndofs ? total : ndofs;
  const int64_t maxblocks = (maxrows + kGatherThreads - 1) / kGatherThreads;
  h->off_rows = kGatherHeaderBytes;
  h->off_len = h->off_rows + gather_align(maxrows * 4);
  h->off_base = h->off_len + gather_align(maxrows);
  h->off_entries = h->off_base + gather_align((maxblocks + 1) * 4);
  h->bytes = h->off_entries + gather_align(total * 4);
}

// R rows per thread (row r0 + k * 256, k < R: every access of a wave stays contiguous).  One row per thread leaves the
// kernel latency-bound: three dependent loads (len -> entry -> detJ) with ~40 bytes in flight per thread is 3.5 TB/s at
// full occupancy (0.117 ms at config 3); R rows per thread issue R independent chains (two rows: 0.100-0.104 ms, 56 VGPRs).
// What was measured and not kept (profiles/r04t_ab_mass_gather.log): a workgroup walking over several blocks and issuing
// the first-level loads of its next block before chasing the entries of the current one -- slower (0.110-0.117 ms: 74
// VGPRs, six waves per SIMD); non-temporal loads of the index streams -- slower (0.120: the per-lane strided entry reads
// live on L1 re-use); the ids of a batch in one 8 / 16-byte load per row -- no change (0.1056; fp32 -3 %).  HBM traffic is 1.06 x the algorithmic bytes already (r04t_mass_gather_counters.json); an ablation
// prices the parts: without the gather of the entity constant 0.097, without the entry indirection (detJ read in row
// order, 79 MB fewer) 0.084, without both 0.076 -- the kernel pays for its vector-memory instructions, not for bytes.
// STATIC (opt-in: the caller declared detJ constant across applies, fus_mass_gather_static_build): detJ is read from a copy
// in ROW order -- detJ_sorted[beg + j], the same contiguous stream the entry ids were -- and the entity of an entry from a
// 16-bit offset to its 256-row block's first entity: the entry indirection (one dependent gather per entry, what keeps the
// texture addresser 82 % busy in the default kernel) is gone; the constants are still gathered, so they may change per apply.
struct GatherStatic {
  const void* detJ_sorted;   // T[nent * N], row order
  const uint16_t* ent16;     // entity of entry k = ent_base[block of its row] + ent16[k]
  const int32_t* ent_base;   // int32[nblocks]
};

template <typename T, int NT, bool DENSE, int R, bool STATIC = false>
__global__ void __launch_bounds__(kGatherThreads)
    mass_gather_kernel(const T* __restrict__ x, const T* __restrict__ cc, T* __restrict__ y, const T* __restrict__ detJ,
                       GatherView v, double inv_n, int chunk, int64_t nkb, GatherStatic gs) {
  // consecutive blocks of rows stay on one XCD (workgroups are dealt round-robin to the 8 XCDs): neighbouring rows gather
  // from the same detJ lines, which then hit in that XCD's L2 (natural block order: 0.1065 against 0.1004 ms)
  const int64_t b = (int64_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (b >= nkb) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = b * (kGatherThreads * R) + threadIdx.x;
  int len[R], incl[R];
  int64_t d[R];
  bool live[R];
  T xv[R], acc[R];
  // the row lengths with a clamped INDEX instead of a predicated load, and selected after x and y have been issued: the compiler
  // waits inside the block of ``live ? p[r] : 0`` when the value is used there, one round trip per row of the thread
  uint8_t lraw[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int64_t r = r0 + k * kGatherThreads;
    live[k] = r < v.nrows;
    lraw[k] = v.len[live[k] ? r : v.nrows - 1];
    d[k] = DENSE ? r : (live[k] ? (int64_t)v.rows[r] : 0);
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    xv[k] = live[k] ? ld_stream<NT>(x + d[k]) : T(0);
    acc[k] = live[k] ? ld_stream<NT>(y + d[k]) : T(0);
  }
#pragma unroll
  for (int k = 0; k < R; ++k) len[k] = live[k] ? (int)lraw[k] : 0;
  // exclusive prefix of len over the workgroup's R * 256 rows (row order: k, wave, lane)
#pragma unroll
  for (int k = 0; k < R; ++k) incl[k] = len[k];
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int up = __shfl_up(incl[k], o, 64);
      if (lane >= o) incl[k] += up;
    }
  }
  constexpr int NW = kGatherThreads / 64;
  __shared__ int wsum[R * NW];
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < R; ++k) wsum[k * NW + wave] = incl[k];
  }
  __syncthreads();
  int64_t beg[R];
  {
    int run = 0;
    const int64_t base = (int64_t)v.base[b * R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        if (w == wave) beg[k] = base + run + incl[k] - len[k];
        run += wsum[k * NW + w];
      }
    }
  }
  // Entries in batches whose loads are all independent (entry ids, then detJ and the constants), predicated per lane; the
  // sums stay in ascending entry order.  84 % of the dofs of a hexahedral mesh have at most two entries, edge dofs four,
  // vertex dofs eight: a wave goes on to the next batch only if one of its lanes needs it.
  int maxlen = len[0];
#pragma unroll
  for (int k = 1; k < R; ++k) maxlen = len[k] > maxlen ? len[k] : maxlen;
  int32_t ebase[R];
  if constexpr (STATIC) {
#pragma unroll
    for (int k = 0; k < R; ++k) ebase[k] = gs.ent_base[b * R + k];  // row r0 + k * 256 lies in 256-row block b * R + k
  }
  // An (empty) unconditional use of the entry ids between their loads and the loads they address: every id is loaded in a predicated
  // block of its own, and without a wait that ALL paths see the compiler waits again after the first dependent block -- for the first
  // detJ / constant pair, one round trip before the others are issued.
  auto pin_entries = [](auto& e) {
#pragma unroll
    for (auto& row : e)
#pragma unroll
      for (auto& v : row) asm volatile("" : "+v"(v));
  };
  auto batch = [&](auto bc, int j0) {
    constexpr int B = decltype(bc)::value;
    uint32_t e[R][B];
    T dv[R][B], cv[R][B];
    if constexpr (STATIC) {
      const T* ds = static_cast<const T*>(gs.detJ_sorted);
#pragma unroll
      for (int k = 0; k < R; ++k)
#pragma unroll
        for (int q = 0; q < B; ++q) {
          const bool on = j0 + q < len[k];
          e[k][q] = on ? (uint32_t)gs.ent16[beg[k] + j0 + q] : 0u;
          dv[k][q] = on ? ds[beg[k] + j0 + q] : T(0);
        }
      pin_entries(e);
#pragma unroll
      for (int k = 0; k < R; ++k)
#pragma unroll
        for (int q = 0; q < B; ++q) cv[k][q] = j0 + q < len[k] ? cc[ebase[k] + (int32_t)e[k][q]] : T(0);
    } else {
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int q = 0; q < B; ++q) e[k][q] = j0 + q < len[k] ? (uint32_t)v.entries[beg[k] + j0 + q] : 0u;
    pin_entries(e);
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const bool on = j0 + q < len[k];
        dv[k][q] = on ? detJ[e[k][q]] : T(0);
        cv[k][q] = on ? cc[(int)(((double)e[k][q] + 0.5) * inv_n)] : T(0);  // e / N, exact for e < 2^31, N <= 2^11
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int q = 0; q < B; ++q)
        if (j0 + q < len[k]) acc[k] += xv[k] * dv[k][q] * cv[k][q];
  };
  batch(std::integral_constant<int, 2>{}, 0);
  if (__any(maxlen > 2)) batch(std::integral_constant<int, 2>{}, 2);
  for (int j0 = 4; __any(j0 < maxlen); j0 += 4) batch(std::integral_constant<int, 4>{}, j0);
#pragma unroll
  for (int k = 0; k < R; ++k)
    if (live[k]) st_stream<NT>(y + d[k], acc[k]);
}

// ---- plan build (device): stable sort of the entries by dof, run lengths, per-block bases
__global__ void gather_iota_kernel(int32_t* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)i;
}
// sorted keys -> row starts: flag[i] = 1 where a new dof begins
__global__ void gather_flag_kernel(const int32_t* keys, int32_t* flag, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}
// rowid[i] = inclusive scan of flag - 1; at a row start: rows[rowid] = key, start[rowid] = i
__global__ void gather_rows_kernel(const int32_t* keys, const int32_t* flag, const int32_t* rowid_incl, int32_t* rows,
                                   int32_t* start, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) {
    const int32_t r = rowid_incl[i] - 1;
    rows[r] = keys[i];
    start[r] = (int32_t)i;
  }
}
// len[r] = start[r + 1] - start[r]; base[b] = start[b * 256]; stats: [0] max len, [1] 1 if rows != 0..nrows-1
__global__ void gather_len_kernel(const int32_t* start, const int32_t* rows, uint8_t* len, int32_t* base, int64_t nrows,
                                  int64_t total, int64_t nblocks, int32_t* stats) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int32_t mx = 0;
  bool sparse = false;
  if (r < nrows) {
    const int64_t nxt = r + 1 < nrows ? start[r + 1] : total;
    const int64_t l = nxt - start[r];
    len[r] = (uint8_t)(l > 255 ? 255 : l);
    mx = l > 255 ? (1 << 30) : (int32_t)l;
    sparse = rows[r] != (int32_t)r;
    if (r % kGatherThreads == 0) base[r / kGatherThreads] = start[r];
  }
  if (r == 0) base[nblocks] = (int32_t)total;
  // one atomic per wave (10 M same-address atomics took 1.8 ms of set-up at config 3)
  for (int o = 32; o > 0; o >>= 1) {
    const int32_t other = __shfl_down(mx, o, 64);
    mx = other > mx ? other : mx;
  }
  const bool any_sparse = __any(sparse);
  if ((threadIdx.x & 63) == 0) {
    if (mx > 0) atomicMax(stats + 0, mx);
    if (any_sparse) atomicMax(stats + 1, 1);
  }
}

// keys of a ROW SUBSET: dofs whose mark is not ``want`` get the sentinel key ``ndofs`` (sorted behind every real dof and dropped);
// invalid dofmap values stay invalid (negative, or beyond the sentinel) so that the range check of the builder still sees them
__global__ void gather_subset_keys_kernel(const int32_t* dofmap, const uint8_t* row_set, int want, int64_t ndofs, int32_t* keys,
                                          int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t d = dofmap[i];
  int32_t k;
  if (d < 0) k = d;
  else if ((int64_t)d >= ndofs) k = 0x7fffffff;
  else k = (row_set[d] == (uint8_t)want) ? d : (int32_t)ndofs;
  keys[i] = k;
}

// Builds the plan in ``ws`` (>= gather_layout(...).bytes).  ``row_set`` != nullptr: a plan of the rows (dofs) d with
// row_set[d] == want only -- the partitioned apply splits the touched dofs into the rows no exchange touches and the rest
// (scatterer.HaloApply), every row still sums ALL its entries.  Scratch: four int32 arrays of nent * N entries and hipCUB's
// temporaries, carved from ONE allocation made and released here (set-up path, once per dofmap: 25 ms at config 3).
// *bad = 1: a dofmap value outside [0, ndofs) or a dof with more than 255 entries (nothing usable was built)
inline hipError_t gather_plan_build(const int32_t* dofmap, int N, int64_t nent, int64_t ndofs, void* ws, hipStream_t stream,
                                    GatherHeader* out, int* bad, const uint8_t* row_set = nullptr, int want = 0) {
  GatherHeader h{};
  *bad = 0;
  gather_layout(nent, N, ndofs, &h);
  const int64_t total = nent * N;
  char* w = static_cast<char*>(ws);
  int32_t* rows = reinterpret_cast<int32_t*>(w + h.off_rows);
  uint8_t* len = reinterpret_cast<uint8_t*>(w + h.off_len);
  int32_t* base = reinterpret_cast<int32_t*>(w + h.off_base);
  int32_t* entries = reinterpret_cast<int32_t*>(w + h.off_entries);
  h.magic = kGatherMagic;
  h.nent = nent;
  h.N = N;
  if (total == 0) {
    h.nrows = h.nblocks = h.max_len = 0;
    h.dense = 1;
    *out = h;
    const hipError_t e0 = hipMemcpyAsync(ws, &h, sizeof h, hipMemcpyHostToDevice, stream);
    return e0 != hipSuccess ? e0 : hipStreamSynchronize(stream);
  }
  size_t sort_bytes = 0, scan_bytes = 0;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const int32_t*)nullptr, (int32_t*)nullptr,
                                                    (const int32_t*)nullptr, (int32_t*)nullptr, (int)total, 0, 32, stream);
  if (e != hipSuccess) return e;
  e = hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int)total, stream);
  if (e != hipSuccess) return e;
  const int64_t arr = gather_align(total * 4);
  const int64_t tmp_bytes = gather_align((int64_t)(sort_bytes > scan_bytes ? sort_bytes : scan_bytes));
  char* scratch = nullptr;
  const int narr = row_set ? 5 : 4;
  e = hipMalloc(&scratch, narr * arr + tmp_bytes + 256);
  if (e != hipSuccess) return e;
  int32_t* iota = reinterpret_cast<int32_t*>(scratch);
  int32_t* keys = reinterpret_cast<int32_t*>(scratch + arr);
  int32_t* flag = reinterpret_cast<int32_t*>(scratch + 2 * arr);  // later: start[]
  int32_t* rowid = reinterpret_cast<int32_t*>(scratch + 3 * arr);
  int32_t* keys_in = row_set ? reinterpret_cast<int32_t*>(scratch + 4 * arr) : nullptr;
  void* tmp = scratch + narr * arr;
  int32_t* stats = reinterpret_cast<int32_t*>(scratch + narr * arr + tmp_bytes);
  const int T = 256;
  const unsigned gb = (unsigned)((total + T - 1) / T);
  hipLaunchKernelGGL(gather_iota_kernel, dim3(gb), dim3(T), 0, stream, iota, total);
  if (row_set) hipLaunchKernelGGL(gather_subset_keys_kernel, dim3(gb), dim3(T), 0, stream, dofmap, row_set, want, ndofs, keys_in, total);
  size_t sb = sort_bytes;
  e = hipcub::DeviceRadixSort::SortPairs(tmp, sb, row_set ? (const int32_t*)keys_in : dofmap, keys, (const int32_t*)iota, entries, (int)total, 0,
                                         32, stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(gather_flag_kernel, dim3(gb), dim3(T), 0, stream, keys, flag, total);
    size_t cb = scan_bytes;
    e = hipcub::DeviceScan::InclusiveSum(tmp, cb, (const int32_t*)flag, rowid, (int)total, stream);
  }
  int32_t nrows32 = 0, kmin = 0, kmax = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&nrows32, rowid + (total - 1), 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&kmin, keys, 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&kmax, keys + (total - 1), 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  // a row subset: the largest legal key is the sentinel ``ndofs`` itself (the dofs of the other subset)
  if (e == hipSuccess && (kmin < 0 || (int64_t)kmax >= ndofs + (row_set ? 1 : 0))) *bad = 1;
  int32_t st[2] = {0, 0};
  if (e == hipSuccess && !*bad) {
    int64_t nrows = nrows32;
    int64_t used = total;  // entries of the kept rows
    int32_t* start = iota;  // iota is no longer needed
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gb), dim3(T), 0, stream, keys, flag, rowid, rows, start, total);
    if (row_set && (int64_t)kmax == ndofs) {  // the last row is the sentinel's: drop it
      nrows -= 1;
      int32_t s32 = 0;
      e = hipMemcpyAsync(&s32, start + nrows, 4, hipMemcpyDeviceToHost, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      used = s32;
    }
    const int64_t nblocks = (nrows + kGatherThreads - 1) / kGatherThreads;
    if (e == hipSuccess) e = hipMemsetAsync(stats, 0, 8, stream);
    if (e == hipSuccess && nrows > 0) {
      hipLaunchKernelGGL(gather_len_kernel, dim3((unsigned)((nrows + T - 1) / T)), dim3(T), 0, stream, start, rows, len, base,
                         nrows, used, nblocks, stats);
      e = hipMemcpyAsync(st, stats, 8, hipMemcpyDeviceToHost, stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    h.nrows = nrows;
    h.nblocks = nblocks;
    h.max_len = st[0];
    h.dense = st[1] ? 0 : 1;
    if (st[0] > 255) *bad = 1;
    if (e == hipSuccess) e = hipMemcpyAsync(ws, &h, sizeof h, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    *out = h;
  }
  (void)hipFree(scratch);
  if (e == hipSuccess) e = hipGetLastError();
  return e;
}

// ---- static companion of a plan (opt-in): detJ in row order + 16-bit entity offsets per 256-row block
// pass 1: per 256-row block, the smallest entity among its entries (entries of a block are the contiguous range base[b] .. base[b + 1])
__global__ void __launch_bounds__(256) gather_static_base_kernel(GatherView v, int N, int32_t* ent_base, int32_t* span_max) {
  const int64_t b = blockIdx.x;
  const int64_t k0 = v.base[b], k1 = v.base[b + 1];
  int32_t lo = 0x7fffffff, hi = 0;
  for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
    const int32_t ent = v.entries[k] / N;
    lo = ent < lo ? ent : lo;
    hi = ent > hi ? ent : hi;
  }
  __shared__ int32_t slo[256], shi[256];
  slo[threadIdx.x] = lo;
  shi[threadIdx.x] = hi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      slo[threadIdx.x] = slo[threadIdx.x + o] < slo[threadIdx.x] ? slo[threadIdx.x + o] : slo[threadIdx.x];
      shi[threadIdx.x] = shi[threadIdx.x + o] > shi[threadIdx.x] ? shi[threadIdx.x + o] : shi[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int32_t base = k1 > k0 ? slo[0] : 0;
    ent_base[b] = base;
    if (k1 > k0) atomicMax(span_max, shi[0] - base);
  }
}
// pass 2: detJ_sorted[k] = detJ[entries[k]], ent16[k] = entries[k] / N - ent_base[block of k]
template <typename T>
__global__ void __launch_bounds__(256) gather_static_fill_kernel(GatherView v, int N, const T* __restrict__ detJ, const int32_t* __restrict__ ent_base,
                                                                 T* __restrict__ detJ_sorted, uint16_t* __restrict__ ent16) {
  const int64_t b = blockIdx.x;
  const int64_t k0 = v.base[b], k1 = v.base[b + 1];
  const int32_t base = ent_base[b];
  for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
    const int32_t e = v.entries[k];
    detJ_sorted[k] = detJ[e];
    ent16[k] = (uint16_t)(e / N - base);
  }
}

// layout of the static companion: [detJ_sorted: total * sizeof(T)] [ent16: total * 2] [ent_base: nblocks * 4], each 256-aligned
inline int64_t gather_static_bytes(int64_t nent, int N, int64_t ndofs, int elem_bytes) {
  const int64_t total = nent * N;
  const int64_t maxrows = total < ndofs ? total : ndofs;
  const int64_t maxblocks = (maxrows + kGatherThreads - 1) / kGatherThreads;
  return gather_align(total * elem_bytes) + gather_align(total * 2) + gather_align((maxblocks + 1) * 4) + 256;
}
inline GatherView gather_view_of(const void* ws, const GatherHeader& h) {
  const char* w = static_cast<const char*>(ws);
  return GatherView{h.nrows, h.nblocks, h.dense ? nullptr : reinterpret_cast<const int32_t*>(w + h.off_rows),
                    reinterpret_cast<const uint8_t*>(w + h.off_len), reinterpret_cast<const int32_t*>(w + h.off_base),
                    reinterpret_cast<const int32_t*>(w + h.off_entries)};
}
inline GatherStatic gather_static_of(void* sws, const GatherHeader& h, int elem_bytes) {
  char* w = static_cast<char*>(sws);
  const int64_t total = h.nent * h.N;
  char* p_ent16 = w + gather_align(total * elem_bytes);
  char* p_base = p_ent16 + gather_align(total * 2);
  return GatherStatic{w, reinterpret_cast<const uint16_t*>(p_ent16), reinterpret_cast<const int32_t*>(p_base)};
}
// *too_wide = 1: some 256-row block spans more than 65 535 entities (a numbering without locality): no static companion
template <typename T>
inline hipError_t gather_static_build(const void* ws, const GatherHeader& h, const T* detJ, void* sws, hipStream_t stream, int* too_wide) {
  *too_wide = 0;
  if (h.nrows == 0) return hipSuccess;
  const GatherView v = gather_view_of(ws, h);
  GatherStatic gs = gather_static_of(sws, h, (int)sizeof(T));
  int32_t* span = const_cast<int32_t*>(gs.ent_base) + h.nblocks;  // one spare word behind the bases
  hipError_t e = hipMemsetAsync(span, 0, 4, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gather_static_base_kernel, dim3((unsigned)h.nblocks), dim3(256), 0, stream, v, (int)h.N, const_cast<int32_t*>(gs.ent_base), span);
  int32_t smax = 0;
  e = hipMemcpyAsync(&smax, span, 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return e;
  if (smax > 65535) {
    *too_wide = 1;
    return hipSuccess;
  }
  hipLaunchKernelGGL((gather_static_fill_kernel<T>), dim3((unsigned)h.nblocks), dim3(256), 0, stream, v, (int)h.N, detJ, gs.ent_base,
                     static_cast<T*>(const_cast<void*>(gs.detJ_sorted)), const_cast<uint16_t*>(gs.ent16));
  return hipGetLastError();
}

template <typename T, int R>
inline hipError_t launch_mass_gather_static_r(const T* x, const T* cc, T* y, const GatherView& v, const GatherHeader& h, const GatherStatic& gs,
                                              hipStream_t stream) {
  const int64_t nkb = (h.nrows + (int64_t)kGatherThreads * R - 1) / ((int64_t)kGatherThreads * R);
  const int chunk = (int)((nkb + 7) / 8);
  const dim3 grid((unsigned)(chunk * 8)), block(kGatherThreads);
  const int nt = vector_stream(h.nrows * (int64_t)sizeof(T)) == 1 ? 1 : 0;
#define FUS_GS(NT_, DENSE_) \
  hipLaunchKernelGGL((mass_gather_kernel<T, NT_, DENSE_, R, true>), grid, block, 0, stream, x, cc, y, (const T*)nullptr, v, 0.0, chunk, nkb, gs)
  if (h.dense) {
    if (nt) FUS_GS(1, true);
    else FUS_GS(0, true);
  } else {
    if (nt) FUS_GS(1, false);
    else FUS_GS(0, false);
  }
#undef FUS_GS
  return hipGetLastError();
}
template <typename T>
inline hipError_t launch_mass_gather_static(const T* x, const T* cc, T* y, const void* ws, const GatherHeader& h, void* sws, hipStream_t stream,
                                            int variant = 0) {
  if (h.nrows == 0) return hipSuccess;
  const GatherView v = gather_view_of(ws, h);
  const GatherStatic gs = gather_static_of(sws, h, (int)sizeof(T));
  int rows_per_thread = variant;
  if (rows_per_thread != 1 && rows_per_thread != 2 && rows_per_thread != 4)
    rows_per_thread = h.nrows < (1 << 19) ? 1 : ((sizeof(T) == 4 && h.nrows >= (1 << 22)) ? 4 : 2);
  switch (rows_per_thread) {
    case 1: return launch_mass_gather_static_r<T, 1>(x, cc, y, v, h, gs, stream);
    case 4: return launch_mass_gather_static_r<T, 4>(x, cc, y, v, h, gs, stream);
    default: return launch_mass_gather_static_r<T, 2>(x, cc, y, v, h, gs, stream);
  }
}

template <typename T, int R>
inline hipError_t launch_mass_gather_r(const T* x, const T* cc, T* y, const T* detJ, const GatherView& v, const GatherHeader& h,
                                       hipStream_t stream) {
  const int64_t nkb = (h.nrows + (int64_t)kGatherThreads * R - 1) / ((int64_t)kGatherThreads * R);
  const int chunk = (int)((nkb + 7) / 8);  // blocks of one XCD
  const dim3 grid((unsigned)(chunk * 8)), block(kGatherThreads);
  const double inv_n = 1.0 / (double)h.N;
  const int nt = vector_stream(h.nrows * (int64_t)sizeof(T)) == 1 ? 1 : 0;
  if (h.dense) {
    if (nt)
      hipLaunchKernelGGL((mass_gather_kernel<T, 1, true, R>), grid, block, 0, stream, x, cc, y, detJ, v, inv_n, chunk, nkb, GatherStatic{nullptr, nullptr, nullptr});
    else
      hipLaunchKernelGGL((mass_gather_kernel<T, 0, true, R>), grid, block, 0, stream, x, cc, y, detJ, v, inv_n, chunk, nkb, GatherStatic{nullptr, nullptr, nullptr});
  } else {
    if (nt)
      hipLaunchKernelGGL((mass_gather_kernel<T, 1, false, R>), grid, block, 0, stream, x, cc, y, detJ, v, inv_n, chunk, nkb, GatherStatic{nullptr, nullptr, nullptr});
    else
      hipLaunchKernelGGL((mass_gather_kernel<T, 0, false, R>), grid, block, 0, stream, x, cc, y, detJ, v, inv_n, chunk, nkb, GatherStatic{nullptr, nullptr, nullptr});
  }
  return hipGetLastError();
}

// ``variant`` (FUS_TUNE_MASS_VARIANT): rows per thread (1, 2, 4; anything else: chosen by size and type)
template <typename T>
inline hipError_t launch_mass_gather(const T* x, const T* cc, T* y, const T* detJ, const void* ws, const GatherHeader& h,
                                     hipStream_t stream, int variant = 0) {
  if (h.nrows == 0) return hipSuccess;
  const char* w = static_cast<const char*>(ws);
  const GatherView v{h.nrows, h.nblocks, h.dense ? nullptr : reinterpret_cast<const int32_t*>(w + h.off_rows),
                     reinterpret_cast<const uint8_t*>(w + h.off_len), reinterpret_cast<const int32_t*>(w + h.off_base),
                     reinterpret_cast<const int32_t*>(w + h.off_entries)};
  int rows_per_thread = variant;
  if (rows_per_thread != 1 && rows_per_thread != 2 && rows_per_thread != 4) {
    // measured (profiles/r04t_ab_mass_gather.log): below ~0.5 M dofs one row per thread fills the chip best; fp64 two rows
    // (0.100 ms at config 3, four rows the same, one row 0.117); fp32 four rows from a few M dofs (0.070 against 0.075)
    rows_per_thread = h.nrows < (1 << 19) ? 1 : ((sizeof(T) == 4 && h.nrows >= (1 << 22)) ? 4 : 2);
  }
  switch (rows_per_thread) {
    case 1: return launch_mass_gather_r<T, 1>(x, cc, y, detJ, v, h, stream);
    case 4: return launch_mass_gather_r<T, 4>(x, cc, y, detJ, v, h, stream);
    default: return launch_mass_gather_r<T, 2>(x, cc, y, detJ, v, h, stream);
  }
}

}  // namespace fus
