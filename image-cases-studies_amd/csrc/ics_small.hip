// ics_small.hip -- the inner iterations of one outer iteration of the Richardson-Lucy loop as ONE cooperative launch, for frames small
// enough that every (tile, channel) of the u-frame has a compute unit of its own (lib/deconvolution.pyx:473-589; the blind phase of
// deconvolve.py:277-286 runs on a 255 x 255 window at every pyramid level).
//
// At 255 x 255 / 15 x 15 the multi-launch path spends 54 us in six kernel bodies and 12 us between them per inner iteration, and the bodies
// are start-up latency: each re-stages its operands from L2.  Here a workgroup = one T x T tile of one channel (T = 32; 64 is built but not
// routed, ics_small_plan; <= 85 tiles x 3 channels <= 256 compute units), 512 threads, and its operands live in LDS for the whole launch:
//     U   u over the tile +- 2 pad    (refilled from global memory after every update: the halo belongs to the neighbours)
//     F   the image over the tile +- pad, UT the majoriser over the tile (both constant for the launch)
//     E   the residual over the tile +- pad: A1 + A2 are evaluated on the halo too (2x the products at 32 / 15) -- cheaper than a fifth barrier
// and an inner iteration is (pyx line numbers as in ics_conv.hip / ics_kernels.hip)
//     A1+A2  E = conv(U, rot180 psf) - F on tile +- pad          A3  G = conv(E, psf) on the tile; max |lambd G + (u - ut)/2|, max u -> the tile's slot
//     -- grid barrier --  A5...A10 on the tile, u -> global      -- grid barrier --  U refilled
//     blind: A11 e' on the tile, A13 the tile's share of the PSF gradient -> global   -- grid barrier --   the shares summed, three taps per
//     workgroup, in tile order   -- grid barrier --   A14...A17 by every workgroup for itself (the PSF step is 3 K^2 values)
// Convolutions are fp32 FMAs in a fixed order: a thread owns 16 consecutive outputs of one row and a group of kernel rows (all kernel-row
// groups of a row together fill the 512 threads), partial sums meet in LDS in group order.
//
// Across workgroups only four things travel, all through agent-scope relaxed atomics (coherent accesses; a __threadfence() either side of a
// barrier costs 20 us on this part -- L2 write-back and invalidate -- against 1.2 us for the accesses, tools/ubench_grid_barrier.hip): the
// updated u, the tiles' step-size maxima, the gradient shares and the summed gradient.  The barrier is sixteen counters (workgroup w arrives on
// counter w % 16), all watched by one wave-wide load per workgroup, monotone over the run (grid_barrier below).
#include "ics_common.h"
#include "ics_kernels.h"

namespace {


__device__ __forceinline__ uint32_t key_of(float f) {
  if (f != f) return 0xFFC00000u;   // canonical NaN: propagates through the integer max like np.amax
  return ics_f2key(f);
}
__device__ __forceinline__ float ld_coh(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_coh(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// i / d for i < 2^32 / d with m = 2^32 / d + 1 (ics_small_plan): one v_mul_hi_u32 instead of the ~35 instructions of an integer division
__device__ __forceinline__ int udiv(int i, uint32_t m) { return (int)__umulhi((uint32_t)i, m); }

// sum of the partial sums p[0], p[stride], ... (n <= MAXG of them) in that order, all of them requested before the first add
template <int MAXG>
__device__ __forceinline__ float sum_partials(const float* __restrict__ p, int stride, int n) {
  float v[MAXG];
#pragma unroll
  for (int g = 0; g < MAXG; ++g) v[g] = p[(g < n ? g : 0) * stride];
  float s = v[0];
#pragma unroll
  for (int g = 1; g < MAXG; ++g) s = g < n ? __fadd_rn(s, v[g]) : s;
  return s;
}

// every workgroup arrives once per generation; `gen` counts from the start of the run (the counters are zeroed with the job's state).
// Arrival = one add on the counter of the workgroup's group (16 groups: 15 adds queue behind each other, not 242); everyone then watches all 16
// counters with one wave-wide load -- no second level to climb (a two-level counter cost a round trip more: 3.3 -> 2 us)
__device__ __forceinline__ void grid_barrier(unsigned long long* bar, unsigned long long gen, int nwg) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's coherent stores have been acknowledged before anyone is told
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (lane == 0) __hip_atomic_fetch_add(bar + 16 * (blockIdx.x & (ICS_SMALL_BAR_GROUPS - 1)), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int gsize = lane < ICS_SMALL_BAR_GROUPS ? (nwg - lane + ICS_SMALL_BAR_GROUPS - 1) / ICS_SMALL_BAR_GROUPS : 0;   // workgroups w with w % 16 == lane
    const unsigned long long target = (gen + 1ull) * (unsigned long long)(gsize > 0 ? gsize : 0);
    while (true) {
      const unsigned long long v = gsize > 0 ? __hip_atomic_load(bar + 16 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
      if (__all(v >= target)) break;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

// out[y][x] = sum_a sum_b w[a][b] in[y + a][x + b] for the kernel rows of one group, 16 outputs per thread; jobs = (row group, column chunk,
// row) with the row fastest (odd pitches: consecutive lanes read consecutive banks)
template <int K, int CW>
__device__ __forceinline__ void conv_row(float (&acc)[CW], const float (&wr)[K], const float (&x)[CW + K - 1]) {
#pragma unroll
  for (int b = 0; b < K; ++b) {
#pragma unroll
    for (int i = 0; i < CW; ++i) acc[i] = __builtin_fmaf(wr[b], x[i + b], acc[i]);
    // the eight packed products of a tap are independent; left to itself the scheduler walks the diagonals (one accumulator three times in a row)
    // and a wave with one mate per SIMD waits out every dependent issue: 1600 clocks per kernel row instead of 600
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int K, int CW>
__device__ __forceinline__ void conv_load(float (&wr)[K], float (&x)[CW + K - 1], const float* __restrict__ wrow, const float* __restrict__ row) {
#pragma unroll
  for (int b = 0; b < K; ++b) wr[b] = wrow[b];
#pragma unroll
  for (int i = 0; i < CW + K - 1; ++i) x[i] = row[i];
}
template <int K, int NT, int CW>
__device__ __forceinline__ void conv_jobs(const float* __restrict__ in, int ip, const float* __restrict__ w, float* __restrict__ part,
                                          int R, int chunks, int Cp, int AG, int AGn, uint32_t m_per, uint32_t m_R, int tid) {
  constexpr int KP = K + 1;
  const int per = R * chunks, jobs = AGn * per;
  for (int job = tid; job < jobs; job += NT) {
    const int ag = udiv(job, m_per), rem = job - ag * per, cx = udiv(rem, m_R), y = rem - cx * R, x0 = cx * CW;
    float acc[CW];
#pragma unroll
    for (int i = 0; i < CW; ++i) acc[i] = 0.f;
    const int a0 = ag * AG, a1 = a0 + AG < K ? a0 + AG : K;
    const float* wrow = w + a0 * KP;
    const float* row = in + (y + a0) * ip + x0;
    for (int a = a0; a < a1; ++a, wrow += KP, row += ip) {
      float wr[K], xr[CW + K - 1];
      conv_load<K, CW>(wr, xr, wrow, row);
      conv_row<K, CW>(acc, wr, xr);
    }
    float* po = part + (ag * R + y) * Cp + x0;
#pragma unroll
    for (int i = 0; i < CW; ++i) po[i] = acc[i];
  }
}

// the tile's share of the PSF gradient (pyx:567-571): gk[a][b] = sum_{y, x} E[y][x] U[y + pad - a][x + pad - b] over the tile's pixels.
// jobs = (a, column chunk, row group), K accumulators per thread
template <int K, int NT, int CW>
__device__ __forceinline__ void gradk_jobs(const float* __restrict__ e_own, int pE, const float* __restrict__ sU, int pU, float* __restrict__ part,
                                           int T, int chunks, int YG, int rpy, uint32_t m_per, uint32_t m_YG, int tid) {
  constexpr int pad = K / 2;
  const int per = chunks * YG, jobs = K * per;
  for (int job = tid; job < jobs; job += NT) {
    const int a = udiv(job, m_per), rem = job - a * per, cx = udiv(rem, m_YG), yg = rem - cx * YG, x0 = cx * CW;
    float acc[K];
#pragma unroll
    for (int b = 0; b < K; ++b) acc[b] = 0.f;
    const int y1 = (yg + 1) * rpy < T ? (yg + 1) * rpy : T;
    for (int y = yg * rpy; y < y1; ++y) {
      float ev[CW], uw[CW + K - 1];
      const float* er = e_own + y * pE + x0;
      const float* ur = sU + (y + 3 * pad - a) * pU + x0 + pad;    // U row 0 = frame row ty0 - 2 pad
#pragma unroll
      for (int i = 0; i < CW; ++i) ev[i] = er[i];
#pragma unroll
      for (int i = 0; i < CW + K - 1; ++i) uw[i] = ur[i];
#pragma unroll
      for (int i = 0; i < CW; ++i) {   // (pixel outer, tap inner: K independent accumulators per step)
#pragma unroll
        for (int b = 0; b < K; ++b) acc[b] = __builtin_fmaf(ev[i], uw[i + K - 1 - b], acc[b]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float* po = part + job * K;
#pragma unroll
    for (int b = 0; b < K; ++b) po[b] = acc[b];
  }
}

// (debug switch small_trace: 100 MHz wall-clock stamps of thread 0 at the phase boundaries, 64 per workgroup)
#define ICS_SMALL_STAMP() do { if (A.trace && tid == 0 && nstamp < 62) A.trace[(size_t)wg * 64 + nstamp++] = wall_clock64(); } while (0)

// U := u over the tile +- 2 pad from a frame in global memory, zero outside the u-frame.  Eight loads per thread are in flight before the first
// is stored (COH: agent-scope atomic loads, which the compiler keeps in program order -- one at a time they were seven round trips)
template <bool COH, int NT>
__device__ __forceinline__ void fill_U(float* __restrict__ sU, const float* __restrict__ src, const IcsGeom& G, const IcsSmallPlan& P, int pad, int ty0, int tx0, int c, int tid) {
  const int HU = P.HU, n = HU * HU;
  const ptrdiff_t pitch = G.pitch;
  for (int base = 0; base < n; base += 8 * NT) {
    float v[8];
    int dst[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + k * NT + tid, ii = i < n ? i : n - 1;
      const int y = udiv(ii, P.m_HU), x = ii - y * HU, fy = ty0 - 2 * pad + y, fx = tx0 - 2 * pad + x;
      const bool ok = i < n && fy >= 0 && fy < G.uM && fx >= 0 && fx < G.uN;
      const int cy = fy < 0 ? 0 : (fy >= G.uM ? G.uM - 1 : fy), cx = fx < 0 ? 0 : (fx >= G.uN ? G.uN - 1 : fx);
      const float* p = src + cy * pitch + 3 * cx + c;
      const float val = COH ? ld_coh(p) : *p;
      v[k] = ok ? val : 0.f;
      dst[k] = i < n ? y * P.pU + x : -1;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) if (dst[k] >= 0) sU[dst[k]] = v[k];
  }
}

template <int K, int NT>
__global__ __launch_bounds__(NT) void k_small_iter(IcsSmallArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ uint32_t skeys[8];
  __shared__ float ssum[4];
  constexpr int pad = K / 2, KP = K + 1, KK = K * K, n3 = 3 * KK, CW = ics_small_cw(K);
  const IcsGeom& G = A.g;
  const IcsSmallPlan& P = A.plan;
  const int tid = threadIdx.x, wg = blockIdx.x;
  const int c = wg % 3, t = wg / 3, tyi = t / P.tiles_x, txi = t - tyi * P.tiles_x;
  const int T = P.T, ty0 = tyi * T, tx0 = txi * T;          // tile origin in u-frame coordinates
  const int ntiles = P.tiles_x * P.tiles_y;
  float* sU = lds + P.off_U; float* sE = lds + P.off_E; float* sF = lds + P.off_F; float* sUT = lds + P.off_UT; float* sG = lds + P.off_G;
  float* sP = lds + P.off_P; float* sW1 = lds + P.off_W; float* sW2 = sW1 + K * KP; float* sPSF = lds + P.off_PSF;
  const int HU = P.HU, pU = P.pU, H1 = P.H1, pE = P.pE, pT = P.pT;
  const ptrdiff_t pitch = G.pitch;
  unsigned long long gen = A.bar_gen;
  int frozen = A.blind ? *A.frozen : 0;
  int nstamp = 0;
  ICS_SMALL_STAMP();
  const unsigned long long clk0 = A.trace ? clock64() : 0ull, wall0 = A.trace ? wall_clock64() : 0ull;

  // ---- operands of the whole launch ----
  fill_U<false, NT>(sU, A.u_in, G, P, pad, ty0, tx0, c, tid);
  for (int i = tid; i < HU * (pU - HU); i += NT) { const int y = i / (pU - HU), x = HU + i - y * (pU - HU); sU[y * pU + x] = 0.f; }   // (columns a chunk over-reads)
  for (int i = tid; i < H1 * H1; i += NT) {
    const int y = udiv(i, P.m_H1), x = i - y * H1, fy = ty0 - pad + y, fx = tx0 - pad + x;
    const bool in = fy >= pad && fy < pad + G.M && fx >= pad && fx < pad + G.N;
    sF[y * pE + x] = in ? A.f[fy * pitch + 3 * fx + c] : 0.f;
  }
  for (int i = tid; i < H1 * pE; i += NT) sE[i] = 0.f;
  for (int i = tid; i < T * T; i += NT) {
    const int y = udiv(i, P.m_T), x = i - y * T, fy = ty0 + y, fx = tx0 + x;
    sUT[y * pT + x] = (fy < G.uM && fx < G.uN) ? A.u_in[fy * pitch + 3 * fx + c] : 0.f;
  }
  for (int i = tid; i < n3; i += NT) sPSF[i] = A.psf[i];
  if (A.psf_bak && wg == 0) for (int i = tid; i < n3; i += NT) { A.psf_bak[i] = A.psf[i]; A.psf_bak[n3 + i] = A.psf_caller[i]; }   // (what an undone outer iteration restores)
  __syncthreads();
  for (int i = tid; i < K * KP; i += NT) {
    const int a = i / KP, b = i - a * KP;
    sW1[i] = b < K ? sPSF[((K - 1 - a) * K + (K - 1 - b)) * 3 + c] : 0.f;    // A1: rot180(psf)   (pyx:441,589)
    sW2[i] = b < K ? sPSF[(a * K + b) * 3 + c] : 0.f;                        // A3: psf
  }
  __syncthreads();

  ICS_SMALL_STAMP();   // operands staged
  for (int it = 0; it < A.inner; ++it) {
    const bool last = it == A.inner - 1;
    uint32_t* red = A.red + it * ICS_RED_STRIDE;
    // ---- A1 + A2 on the tile +- pad (pyx:477-488) ----
    conv_jobs<K, NT, CW>(sU, pU, sW1, sP, H1, P.chunks1, P.Cp1, P.AG1, P.AGn1, P.m_per1, P.m_H1, tid);
    __syncthreads();
    for (int i = tid; i < H1 * H1; i += NT) {
      const int y = udiv(i, P.m_H1), x = i - y * H1, fy = ty0 - pad + y, fx = tx0 - pad + x;
      const float s = sum_partials<4>(sP + y * P.Cp1 + x, H1 * P.Cp1, P.AGn1);
      const bool in = fy >= pad && fy < pad + G.M && fx >= pad && fx < pad + G.N;
      const float e = in ? __fsub_rn(s, sF[y * pE + x]) : 0.f;
      sE[y * pE + x] = e;
      // (non-blind: the residual the stop test reads is the last inner iteration's, pyx:601,627)
      if (!A.blind && last && in && y >= pad && y < pad + T && x >= pad && x < pad + T) A.e[fy * pitch + 3 * fx + c] = e;
    }
    __syncthreads();
    ICS_SMALL_STAMP();   // A1 + A2
    // ---- A3 on the tile (pyx:490-491), A6 / A7 maxima (pyx:512-524) ----
    conv_jobs<K, NT, CW>(sE, pE, sW2, sP, T, P.chunksT, P.CpT, P.AGT, P.AGnT, P.m_perT, P.m_T, tid);
    if (tid < 8) skeys[tid] = tid == 2 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    uint32_t kg = 0u, ku = 0u;
    for (int i = tid; i < T * T; i += NT) {
      const int y = udiv(i, P.m_T), x = i - y * T, fy = ty0 + y, fx = tx0 + x;
      const float s = sum_partials<8>(sP + y * P.CpT + x, T * P.CpT, P.AGnT);
      sG[y * pT + x] = s;
      if (fy < G.uM && fx < G.uN) {
        const float uv = sU[(y + 2 * pad) * pU + x + 2 * pad];
        const float g6 = __fadd_rn(__fmul_rn(A.lambd, s), __fmul_rn(__fsub_rn(uv, sUT[y * pT + x]), 0.5f));
        const uint32_t k1 = key_of(__builtin_fabsf(g6)), k2 = key_of(uv);
        kg = kg > k1 ? kg : k1; ku = ku > k2 ? ku : k2;
      }
    }
    kg = ics_wave_max_u32(kg); ku = ics_wave_max_u32(ku);
    if ((tid & 63) == 0) { atomicMax(&skeys[0], kg); atomicMax(&skeys[1], ku); }
    __syncthreads();
    // the tile's two maxima go to a slot of their own (one 64-bit word): 81 atomics per address queued for 3 us in front of the barrier
    unsigned long long* keys = A.keys + (size_t)it * P.nwg;
    if (tid == 0) __hip_atomic_store(keys + wg, ((unsigned long long)skeys[0] << 32) | skeys[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ICS_SMALL_STAMP();   // A3
    grid_barrier(A.bar, gen++, P.nwg);
    ICS_SMALL_STAMP();   // barrier
    // ---- A5 ... A10 on the tile (pyx:499-552); the updated u goes to global memory for the neighbours and the statistics ----
    float maxu, maxg;
    {   // every wave for itself: lane = tile of this channel (two per lane beyond 64 tiles), both loads in flight together
      const int lane = tid & 63, l0 = lane < ntiles ? lane : ntiles - 1, l1 = lane + 64 < ntiles ? lane + 64 : ntiles - 1;
      const unsigned long long v0 = __hip_atomic_load(keys + 3 * l0 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long v1 = __hip_atomic_load(keys + 3 * l1 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t k0 = (uint32_t)(v0 >> 32), k1 = (uint32_t)v0;
      const uint32_t q0 = (uint32_t)(v1 >> 32), q1 = (uint32_t)v1;
      k0 = k0 > q0 ? k0 : q0; k1 = k1 > q1 ? k1 : q1;
      k0 = ics_wave_max_u32(k0); k1 = ics_wave_max_u32(k1);
      maxg = ics_key2f(k0); maxu = ics_key2f(k1);
      if (t == 0 && tid == 0) { red[ICS_RED_MAXG + c] = k0; red[ICS_RED_MAXU + c] = k1; }   // (where the multi-launch path leaves them)
    }
    const float dt = __fdiv_rn(__fmul_rn(A.step, maxu), __fadd_rn(maxg, 1e-15f));
    if (t == 0 && tid == 0) { A.scal[ICS_SC_DT + c] = dt; A.scal[ICS_SC_MAXU + c] = maxu; A.scal[ICS_SC_MAXG + c] = maxg; }
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u, knan = 0u;
    for (int i = tid; i < T * T; i += NT) {
      const int y = udiv(i, P.m_T), x = i - y * T, fy = ty0 + y, fx = tx0 + x;
      if (fy >= G.uM || fx >= G.uN) continue;
      const float uv = sU[(y + 2 * pad) * pU + x + 2 * pad], gv = sG[y * pT + x];
      const float g6 = __fadd_rn(__fmul_rn(A.lambd, gv), __fmul_rn(__fsub_rn(uv, sUT[y * pT + x]), 0.5f));
      float un = __fsub_rn(uv, __fmul_rn(dt, g6));
      if (fy >= pad && fy < pad + G.M && fx >= pad && fx < pad + G.N) {
        const float fv = sF[(y + pad) * pE + x + pad];
        const float d = ics_dof_ratio(gv, fv);
        float D = __fmul_rn(d, d);
        if (!A.blind) D = __fdiv_rn(D, A.lambd);
        un = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, D), un), __fmul_rn(D, fv));
        if (last) {
          if (D != D) knan = 1u;
          else { const uint32_t k = ics_f2key(D); kmin = kmin < k ? kmin : k; kmax = kmax > k ? kmax : k; }
        }
      }
      st_coh(A.u_out + fy * pitch + 3 * fx + c, un);
    }
    if (last) {   // DoF keys of the outer iteration (pyx:593): one set of atomics per workgroup
      kmax = ics_wave_max_u32(kmax); knan = ics_wave_max_u32(knan); kmin = ~ics_wave_max_u32(~kmin);
      if ((tid & 63) == 0) { atomicMin(&skeys[2], kmin); atomicMax(&skeys[3], kmax); atomicMax(&skeys[4], knan); }
      __syncthreads();
      if (tid == 0) { atomicMin(A.dofkeys + 0, skeys[2]); atomicMax(A.dofkeys + 1, skeys[3]); if (skeys[4]) atomicOr(A.dofkeys + 2, 1u); }
    }
    ICS_SMALL_STAMP();   // update
    if (!A.blind && last) break;
    grid_barrier(A.bar, gen++, P.nwg);
    ICS_SMALL_STAMP();   // barrier
    // ---- U refilled: the tile's own values and the neighbours' ----
    fill_U<true, NT>(sU, A.u_out, G, P, pad, ty0, tx0, c, tid);
    __syncthreads();
    ICS_SMALL_STAMP();   // refill
    if (!A.blind) continue;
    // ---- A11 on the tile (pyx:557-565) ----
    conv_jobs<K, NT, CW>(sU + pad * pU + pad, pU, sW1, sP, T, P.chunksT, P.CpT, P.AGT, P.AGnT, P.m_perT, P.m_T, tid);
    __syncthreads();
    for (int i = tid; i < T * T; i += NT) {
      const int y = udiv(i, P.m_T), x = i - y * T, fy = ty0 + y, fx = tx0 + x;
      const float s = sum_partials<8>(sP + y * P.CpT + x, T * P.CpT, P.AGnT);
      const bool in = fy >= pad && fy < pad + G.M && fx >= pad && fx < pad + G.N;
      const float e = in ? __fsub_rn(s, sF[(y + pad) * pE + x + pad]) : 0.f;
      sE[(y + pad) * pE + x + pad] = e;
      if (last && in) A.e[fy * pitch + 3 * fx + c] = e;
    }
    __syncthreads();
    ICS_SMALL_STAMP();   // A11
    // ---- A12 + A13: the tile's share, then the shares of all tiles (pyx:567-571) ----
    gradk_jobs<K, NT, CW>(sE + pad * pE + pad, pE, sU, pU, sP, T, P.chunksT, P.YG, P.rpy, P.m_perG, P.m_YG, tid);
    __syncthreads();
    {
      const int per = P.chunksT * P.YG;
      float* share = A.part + (size_t)(c * ntiles + t) * KK;
      for (int l = tid; l < KK; l += NT) {
        const int a = l / K, b = l - a * K;
        float s = 0.f;
        for (int q0 = 0; q0 < per; q0 += 8) {   // (eight reads in flight, the adds in job order)
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = sP[(a * per + (q0 + k < per ? q0 + k : q0)) * K + b];
#pragma unroll
          for (int k = 0; k < 8; ++k) s = q0 + k < per ? __fadd_rn(s, v[k]) : s;
        }
        st_coh(share + l, s);
      }
    }
    ICS_SMALL_STAMP();   // A13
    grid_barrier(A.bar, gen++, P.nwg);
    ICS_SMALL_STAMP();   // barrier
    {   // tap l of channel c: the shares of the tiles in tile order, one wave per tap (lane = tile, then a fixed butterfly)
      const int wv = tid >> 6, lane = tid & 63;
      for (int l = t + ntiles * wv; l < KK; l += ntiles * (NT / 64)) {
        const float* col = A.part + (size_t)c * ntiles * KK + l;
        float s = lane < ntiles ? ld_coh(col + (size_t)lane * KK) : 0.f;
        if (lane + 64 < ntiles) s = __fadd_rn(s, ld_coh(col + (size_t)(lane + 64) * KK));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s = __fadd_rn(s, __shfl_xor(s, off, 64));
        if (lane == 0) st_coh(A.gradk + 3 * l + c, s);
      }
    }
    ICS_SMALL_STAMP();   // shares summed
    grid_barrier(A.bar, gen++, P.nwg);
    ICS_SMALL_STAMP();   // barrier
    // ---- A14 ... A17 (pyx:574-589): every workgroup steps the PSF for itself (k_psf's arithmetic, ics_kernels.hip) ----
    {
      constexpr int NG = (n3 + NT - 1) / NT;
      float gv[NG];
#pragma unroll
      for (int r = 0; r < NG; ++r) { const int i = tid + r * NT; gv[r] = ld_coh(A.gradk + (i < n3 ? i : n3 - 1)); }
      uint32_t kp = 0u, kq = 0u;     // (their slots, skeys[5] and [6], were cleared with the others before A3's epilogue of this inner iteration)
#pragma unroll
      for (int r = 0; r < NG; ++r) {
        const int i = tid + r * NT;
        if (i < n3) { const uint32_t k1 = key_of(sPSF[i]), k2 = key_of(__builtin_fabsf(gv[r])); kp = kp > k1 ? kp : k1; kq = kq > k2 ? kq : k2; }
      }
      kp = ics_wave_max_u32(kp); kq = ics_wave_max_u32(kq);
      if ((tid & 63) == 0) { atomicMax(&skeys[5], kp); atomicMax(&skeys[6], kq); }
      __syncthreads();
      const float maxp = ics_key2f(skeys[5]), maxq = ics_key2f(skeys[6]);
      const float dtpsf = __fdiv_rn(__fmul_rn(__fdiv_rn(A.step, (float)K), maxp), __fadd_rn(maxq, 1e-15f));
      const bool writer = wg == 0;
      if (writer && tid == 0) A.scal[ICS_SC_DTPSF] = dtpsf;
#pragma unroll
      for (int r = 0; r < NG; ++r) {
        const int i = tid + r * NT;
        if (i < n3) {
          float v = __fsub_rn(sPSF[i], __fmul_rn(dtpsf, gv[r]));
          if (writer && !frozen) A.psf_caller[i] = v;
          if (!A.correlation) v = v < 0.f ? 0.f : v;
          sPSF[i] = v;
        }
      }
      __syncthreads();
      if (A.correlation) {
        for (int i = tid; i < KK; i += NT) {
          float m = __fdiv_rn(__fadd_rn(__fadd_rn(sPSF[3 * i], sPSF[3 * i + 1]), sPSF[3 * i + 2]), 3.0f);
          m = m < 0.f ? 0.f : m;
          sPSF[3 * i] = m; sPSF[3 * i + 1] = m; sPSF[3 * i + 2] = m;
        }
        __syncthreads();
      }
      if (tid < 3) {   // sequential float32 sum in the reference's order (pyx:58-64)
        float s = 0.f;
        int i = 0;
        for (; i + 32 <= KK; i += 32) {
          float v[32];
#pragma unroll
          for (int k = 0; k < 32; ++k) v[k] = sPSF[3 * (i + k) + tid];
#pragma unroll
          for (int k = 0; k < 32; ++k) s = __fadd_rn(s, v[k]);
        }
        for (; i < KK; ++i) s = __fadd_rn(s, sPSF[3 * i + tid]);
        ssum[tid] = s;
      }
      __syncthreads();
      const bool detach = A.correlation != 0;
      for (int i = tid; i < n3; i += NT) {   // A16, and this channel's values straight into the two weight tables (their pad column stays zero)
        const int px = i / 3, ch = i - 3 * px;
        const float v = __fdiv_rn(sPSF[i], ssum[ch]);
        sPSF[i] = v;
        if (ch == c) { const int a = px / K, b = px - a * K; sW2[a * KP + b] = v; sW1[(K - 1 - a) * KP + (K - 1 - b)] = v; }
        if (writer) { A.psf[i] = v; if (!frozen && !detach) A.psf_caller[i] = v; }
      }
      if (detach) { frozen = 1; if (writer && tid == 0) *A.frozen = 1; }
    }
    __syncthreads();
    ICS_SMALL_STAMP();   // PSF step
  }
  if (A.trace && tid == 0) { A.trace[(size_t)wg * 64 + 62] = clock64() - clk0; A.trace[(size_t)wg * 64 + 63] = wall_clock64() - wall0; }   // shader clocks per 10 ns tick
}

template <int K, int NT>
hipError_t launch_k(const IcsSmallArgs& a, hipStream_t s) {
  static std::atomic<bool> configured[ICS_MAX_DEVICES];
  const size_t lds = (size_t)a.plan.lds_floats * sizeof(float);
  hipError_t e = ics_configure_lds(configured, ics_current_device(), k_small_iter<K, NT>, lds);
  if (e != hipSuccess) return e;
  IcsSmallArgs copy = a;
  void* args[] = {&copy};
  e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_small_iter<K, NT>), dim3((unsigned)a.plan.nwg), dim3(NT), args, (unsigned)lds, s);
  if (e != hipSuccess) { (void)hipGetLastError(); return e; }
  return hipGetLastError();
}

}  // namespace

// The shape of the launch for a frame, or false: the tiles x 3 channels must each find a compute unit and the operands must fit its LDS
bool ics_small_plan(const IcsGeom& g, int cus, IcsSmallPlan* out, bool allow64) {
  const int K = g.K, pad = g.pad;
  if (K < 3 || K > ICS_SMALL_MAX_K || !(K & 1)) return false;
  if (cus <= 0) cus = ics_device_cus(ics_current_device());
  const int NT = ics_small_threads(K), CW = ics_small_cw(K);
  for (int T = 32; T <= (allow64 ? 64 : 32); T *= 2) {
    IcsSmallPlan p;
    memset(&p, 0, sizeof p);
    p.T = T; p.tiles_x = (g.uN + T - 1) / T; p.tiles_y = (g.uM + T - 1) / T; p.nwg = 3 * p.tiles_x * p.tiles_y;
    if (p.nwg > cus || p.tiles_x * p.tiles_y > 128) continue;    // (the gradient's shares are summed by one wave: two per lane at most)
    auto odd = [](int v) { return v | 1; };
    p.HU = T + 4 * pad; p.H1 = T + 2 * pad;
    p.chunks1 = (p.H1 + CW - 1) / CW; p.chunksT = T / CW;
    p.Cp1 = odd(p.chunks1 * CW); p.CpT = odd(T);
    const int needU = p.chunks1 * CW + K - 1;
    p.pU = odd(p.HU > needU ? p.HU : needU); p.pE = odd(p.H1); p.pT = odd(T);
    auto groups = [&](int R, int chunks, int cap, int* AG, int* AGn) {
      int n = NT / (R * chunks);
      n = n < 1 ? 1 : (n > K ? K : n);
      n = n > cap ? cap : n;                   // (sum_partials<cap>)
      *AG = (K + n - 1) / n; *AGn = (K + *AG - 1) / *AG;
    };
    groups(p.H1, p.chunks1, 4, &p.AG1, &p.AGn1);
    groups(T, p.chunksT, 8, &p.AGT, &p.AGnT);
    int yg = NT / (K * p.chunksT);
    yg = yg < 1 ? 1 : (yg > T ? T : yg);
    p.rpy = (T + yg - 1) / yg; p.YG = (T + p.rpy - 1) / p.rpy;
    const int p1 = p.AGn1 * p.H1 * p.Cp1, p2 = p.AGnT * T * p.CpT, p3 = K * p.chunksT * p.YG * K;
    const int pP = p1 > p2 ? (p1 > p3 ? p1 : p3) : (p2 > p3 ? p2 : p3);
    int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 3) & ~3; return at; };
    p.off_U = take((p.HU + 1) * p.pU);      // (+1 row: the last chunk of the last row reads on to the pitch's end)
    p.off_E = take((p.H1 + 1) * p.pE); p.off_F = take((p.H1 + 1) * p.pE);
    p.off_UT = take(T * p.pT); p.off_G = take(T * p.pT);
    p.off_P = take(pP); p.off_W = take(2 * K * (K + 1)); p.off_PSF = take(3 * K * K);
    p.lds_floats = o;
    auto magic = [](int d) { return (uint32_t)(0x100000000ull / (unsigned long long)d) + 1u; };
    p.m_HU = magic(p.HU); p.m_H1 = magic(p.H1); p.m_T = magic(T); p.m_per1 = magic(p.H1 * p.chunks1); p.m_perT = magic(T * p.chunksT);
    p.m_perG = magic(p.chunksT * p.YG); p.m_YG = magic(p.YG);
    if ((size_t)o * 4 + 256 > ICS_SMALL_LDS_BYTES) continue;
    *out = p;
    return true;
  }
  return false;
}

hipError_t ics_launch_small_iter(const IcsSmallArgs& a, hipStream_t s) {
  switch (a.g.K) {
#define ICS_SMALL_CASE(k) case k: return launch_k<k, ics_small_threads(k)>(a, s);
    ICS_SMALL_CASE(3) ICS_SMALL_CASE(5) ICS_SMALL_CASE(7) ICS_SMALL_CASE(9) ICS_SMALL_CASE(11) ICS_SMALL_CASE(13) ICS_SMALL_CASE(15)
    ICS_SMALL_CASE(17) ICS_SMALL_CASE(19) ICS_SMALL_CASE(21) ICS_SMALL_CASE(23) ICS_SMALL_CASE(25) ICS_SMALL_CASE(27) ICS_SMALL_CASE(29) ICS_SMALL_CASE(31)
#undef ICS_SMALL_CASE
    default: return hipErrorInvalidValue;
  }
}
