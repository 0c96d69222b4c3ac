/*
 * ORACLE (test infrastructure -- NOT product code).  Included twice by
 * fus_oracle.c with REAL/SUF defined.  Loop-for-loop C restatement of
 *   numba-cpu/sum_factorisation.py:17-48  (transpose)
 *   numba-cpu/sum_factorisation.py:51-95  (contract)
 *   numba-cpu/operators.py:19-68          (mass_operator)
 *   numba-cpu/operators.py:71-227         (stiffness_operator)
 *   numba-cpu/operators.py:230-300        (axpy, copy, fill, pointwise_divide)
 *   cuda/operators.py:261-274             (square)
 *   numba-cpu/scatterer.py:18-75          (pack, unpack_rev, unpack_fwd)
 * (paths relative to the reference checkout).
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* B[offa*a + offb*b + offc*c] = A[a*Nb*Nc + b*Nc + c]   (sum_factorisation.py:43-46) */
static inline __attribute__((always_inline)) void FN(transpose)(
    int Na, int Nb, int Nc, int offa, int offb, int offc, const REAL* A, REAL* B) {
  for (int a = 0; a < Na; ++a)
    for (int b = 0; b < Nb; ++b)
      for (int c = 0; c < Nc; ++c) B[offa * a + offb * b + offc * c] = A[a * Nb * Nc + b * Nc + c];
}

/* C[a*Nd + d] += A[a*Nk + k] * B[k*Nd + d]  (transposeA)  or  A[k*Na + a]   (sum_factorisation.py:84-93) */
static inline __attribute__((always_inline)) void FN(contract)(
    int Nk, int Na, int Nb, int Nc, int transposeA, const REAL* A, const REAL* B, REAL* C) {
  const int Nd = Nb * Nc;
  if (transposeA) {
    for (int k = 0; k < Nk; ++k)
      for (int a = 0; a < Na; ++a)
        for (int d = 0; d < Nd; ++d) C[a * Nd + d] += A[a * Nk + k] * B[k * Nd + d];
  } else {
    for (int k = 0; k < Nk; ++k)
      for (int a = 0; a < Na; ++a)
        for (int d = 0; d < Nd; ++d) C[a * Nd + d] += A[k * Na + a] * B[k * Nd + d];
  }
}

/* One cell of operators.py:159-225.  ``acc`` selects how the result is added to y. */
static inline __attribute__((always_inline)) void FN(stiffness_cell)(
    const int n, const REAL* restrict x, const REAL cell_constant, REAL* restrict y, const REAL* restrict Gc,
    const int32_t* restrict dofs, const REAL* restrict dphi, REAL* restrict w, int use_atomic) {
  const int N = n * n * n;
  REAL *x_ = w, *T1 = w + N, *T2 = w + 2 * N, *T3 = w + 3 * N, *T4 = w + 4 * N;
  REAL *fw0 = w + 5 * N, *fw1 = w + 6 * N, *fw2 = w + 7 * N;
  REAL *y0_ = w + 8 * N, *y1_ = w + 9 * N, *y2_ = w + 10 * N;
  memset(T1, 0, sizeof(REAL) * 7 * N); /* T1..T4, fw0..fw2  (:160-167) */

  for (int i = 0; i < N; ++i) x_[i] = x[dofs[i]]; /* :170-171 */

  FN(contract)(n, n, n, n, 1, dphi, x_, fw0); /* :174-176 */

  FN(transpose)(n, n, n, n, n * n, 1, x_, T1); /* :179-183 */
  FN(contract)(n, n, n, n, 1, dphi, T1, T2);
  FN(transpose)(n, n, n, n, n * n, 1, T2, fw1);

  FN(transpose)(n, n, n, 1, n, n * n, x_, T3); /* :186-190 */
  FN(contract)(n, n, n, n, 1, dphi, T3, T4);
  FN(transpose)(n, n, n, 1, n, n * n, T4, fw2);

  for (int q = 0; q < N; ++q) { /* stiffness_transform :91-119 */
    const REAL* G_ = Gc + 6 * q;
    const REAL w0 = fw0[q], w1 = fw1[q], w2 = fw2[q];
    fw0[q] = cell_constant * (G_[0] * w0 + G_[1] * w1 + G_[2] * w2);
    fw1[q] = cell_constant * (G_[1] * w0 + G_[3] * w1 + G_[4] * w2);
    fw2[q] = cell_constant * (G_[2] * w0 + G_[4] * w1 + G_[5] * w2);
  }

  memset(T1, 0, sizeof(REAL) * 4 * N);  /* :195-198 */
  memset(y0_, 0, sizeof(REAL) * 3 * N); /* :200-202 */

  FN(contract)(n, n, n, n, 0, dphi, fw0, y0_); /* :205-207 */

  FN(transpose)(n, n, n, n, n * n, 1, fw1, T1); /* :210-214 */
  FN(contract)(n, n, n, n, 0, dphi, T1, T2);
  FN(transpose)(n, n, n, n, n * n, 1, T2, y1_);

  FN(transpose)(n, n, n, 1, n, n * n, fw2, T3); /* :217-221 */
  FN(contract)(n, n, n, n, 0, dphi, T3, T4);
  FN(transpose)(n, n, n, 1, n, n * n, T4, y2_);

  if (use_atomic) {
    for (int i = 0; i < N; ++i) {
      const REAL v = y0_[i] + y1_[i] + y2_[i];
#pragma omp atomic
      y[dofs[i]] += v;
    }
  } else {
    for (int i = 0; i < N; ++i) y[dofs[i]] += y0_[i] + y1_[i] + y2_[i]; /* :224-225 */
  }
}

static void FN(stiffness_range)(int P, const REAL* x, const REAL* cc, REAL* y, const REAL* G, const int32_t* dofmap,
                                const REAL* dphi, int64_t c0, int64_t c1, REAL* w, int use_atomic) {
  const int n = P + 1;
  const int64_t N = (int64_t)n * n * n;
#define RUN(NN)                                                                                              \
  for (int64_t c = c0; c < c1; ++c)                                                                          \
    FN(stiffness_cell)(NN, x, cc[c], y, G + c * N * 6, dofmap + c * N, dphi, w, use_atomic);
  switch (n) { /* literal n => the compiler specialises the loop nests, like numba's closure constants */
    case 2: RUN(2) break;
    case 3: RUN(3) break;
    case 4: RUN(4) break;
    case 5: RUN(5) break;
    case 6: RUN(6) break;
    case 7: RUN(7) break;
    case 8: RUN(8) break;
    case 9: RUN(9) break;
    default: RUN(n) break;
  }
#undef RUN
}

/* Serial: exactly what the reference runs (njit without parallel=True). */
int FN(oracle_stiffness_apply)(const REAL* x, const REAL* cell_constants, REAL* y, const REAL* G,
                               const int32_t* dofmap, const REAL* dphi, int P, int64_t ncell) {
  const int n = P + 1;
  REAL* w = (REAL*)malloc(sizeof(REAL) * 11 * n * n * n);
  if (!w) return -1;
  FN(stiffness_range)(P, x, cell_constants, y, G, dofmap, dphi, 0, ncell, w, 0);
  free(w);
  return 0;
}

/* All host cores.  Cells are cut into contiguous chunks (8 per thread); chunks whose dof
 * index ranges overlap get different colours (greedy, in chunk order); one colour class at a
 * time runs in parallel without atomics.  Any cell ordering with locality (e.g. lexicographic
 * boxes) needs ~3 colours; a pathological ordering (> 32 colours) falls back to omp atomics. */
int FN(oracle_stiffness_apply_omp)(const REAL* x, const REAL* cell_constants, REAL* y, const REAL* G,
                                   const int32_t* dofmap, const REAL* dphi, int P, int64_t ncell, int nthreads) {
  const int n = P + 1;
  const int64_t N = (int64_t)n * n * n;
  if (nthreads < 1) nthreads = 1;
  if (ncell <= 0) return 0;
  int nchunk = 8 * nthreads;
  if (nchunk > ncell) nchunk = (int)ncell;
  int32_t* lo = (int32_t*)malloc(sizeof(int32_t) * 3 * nchunk);
  if (!lo) return -1;
  int32_t* hi = lo + nchunk;
  int32_t* colour = lo + 2 * nchunk;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int k = 0; k < nchunk; ++k) {
    int64_t c0 = ncell * k / nchunk, c1 = ncell * (k + 1) / nchunk;
    int32_t mn = INT32_MAX, mx = INT32_MIN;
    for (int64_t i = c0 * N; i < c1 * N; ++i) {
      if (dofmap[i] < mn) mn = dofmap[i];
      if (dofmap[i] > mx) mx = dofmap[i];
    }
    lo[k] = mn;
    hi[k] = mx;
  }
  enum { MAXCOL = 32 };
  int ncol = 0, use_atomic = 0;
  for (int k = 0; k < nchunk && !use_atomic; ++k) {
    uint32_t used = 0;
    for (int j = 0; j < k; ++j)
      if (!(hi[j] < lo[k] || hi[k] < lo[j])) used |= 1u << colour[j];
    int c = 0;
    while (c < MAXCOL && (used >> c & 1u)) ++c;
    if (c >= MAXCOL) use_atomic = 1;
    colour[k] = c;
    if (c + 1 > ncol) ncol = c + 1;
  }
  if (use_atomic) {
    ncol = 1;
    for (int k = 0; k < nchunk; ++k) colour[k] = 0;
  }
  int err = 0;
#pragma omp parallel num_threads(nthreads)
  {
    REAL* w = (REAL*)malloc(sizeof(REAL) * 11 * N);
    if (!w) {
#pragma omp atomic write
      err = -1;
    }
    for (int col = 0; col < ncol; ++col) {
#pragma omp for schedule(dynamic, 1)
      for (int k = 0; k < nchunk; ++k) {
        if (colour[k] != col || !w) continue;
        int64_t c0 = ncell * k / nchunk, c1 = ncell * (k + 1) / nchunk;
        FN(stiffness_range)(P, x, cell_constants, y, G, dofmap, dphi, c0, c1, w, use_atomic);
      }
    }
    free(w);
  }
  free(lo);
  return err;
}

/* operators.py:50-66 */
int FN(oracle_mass_apply)(const REAL* x, const REAL* entity_constants, REAL* y, const REAL* entity_detJ,
                          const int32_t* entity_dofmap, int N, int64_t num_entities) {
  REAL* x_ = (REAL*)malloc(sizeof(REAL) * (N > 0 ? N : 1));
  if (!x_) return -1;
  for (int64_t e = 0; e < num_entities; ++e) {
    const int32_t* d = entity_dofmap + e * N;
    for (int i = 0; i < N; ++i) x_[i] = x[d[i]];
    for (int i = 0; i < N; ++i) x_[i] *= entity_detJ[e * N + i] * entity_constants[e];
    for (int i = 0; i < N; ++i) y[d[i]] += x_[i];
  }
  free(x_);
  return 0;
}

void FN(oracle_axpy)(REAL alpha, const REAL* x, REAL* y, int64_t n) {
  for (int64_t i = 0; i < n; ++i) y[i] = alpha * x[i] + y[i];
}
void FN(oracle_copy)(const REAL* a, REAL* b, int64_t n) {
  for (int64_t i = 0; i < n; ++i) b[i] = a[i];
}
void FN(oracle_fill)(REAL alpha, REAL* x, int64_t n) {
  for (int64_t i = 0; i < n; ++i) x[i] = alpha;
}
void FN(oracle_pointwise_divide)(const REAL* a, const REAL* b, REAL* c, int64_t n) {
  for (int64_t i = 0; i < n; ++i) c[i] = a[i] / b[i];
}
void FN(oracle_square)(const REAL* a, REAL* b, int64_t n) {
  for (int64_t i = 0; i < n; ++i) b[i] = a[i] * a[i];
}
/* scatterer.py:18-75 */
void FN(oracle_pack)(const REAL* in_, REAL* out_, const int64_t* index, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out_[i] = in_[index[i]];
}
void FN(oracle_unpack_rev)(const REAL* in_, REAL* out_, const int64_t* index, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out_[index[i]] += in_[i];
}
void FN(oracle_unpack_fwd)(const REAL* in_, REAL* out_, const int64_t* index, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out_[index[i]] = in_[i];
}

#undef FN
#undef CAT
#undef CAT_
