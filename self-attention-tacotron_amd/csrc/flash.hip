// Fused scaled-dot-product self-attention for gfx950 (reference modules/self_attention.py:45-65,79-86): QK^T -> scale ->
// causal mask -> softmax (fp32, online) -> counter-based dropout on the probabilities -> PV in ONE kernel; the [B*H, T, T]
// score / probability tensors never exist in memory.  Backward = two kernels that recompute P from Q, K and the saved
// row log-sum-exp: dK / dV (one workgroup per key tile, loop over query tiles) and dQ (one per query tile, loop over key
// tiles) - no atomics, every gradient element has exactly one writer.  Head dimension 128 (the decoder's 2 x 128).
//
// Register scheme (v_mfma_f32_16x16x32_bf16, wave64): every product is arranged so that the softmax axis of a wave sits in
// the MFMA *column* (lane & 15):
//   forward / dQ kernel : S^T = K Q^T       C[row = key, col = query]   -> P^T is directly the B operand of  O^T = V^T P^T
//   dK/dV kernel        : S   = Q K^T       C[row = query, col = key]   -> P, dS are directly the B operands of dV^T = dO^T P,
//                                                                            dK^T = Q^T dS
// so probabilities never take a round trip through LDS: a C tile (rows 4g + r of lane group g = lane >> 4) IS a B operand
// once the reduction index is enumerated as  32 ks + 16 (e / 4) + 4 g + e % 4  (element e of lane group g) - the staged A
// operand (the "transposed" LDS image below) simply uses the same enumeration.  Per-query statistics (max, sum, alpha) live
// in the 4 lane groups of a column and are reduced with two cross-lane steps (xor 16, xor 32).
//
// LDS images of a [64 rows][128] fp32 tile (bf16 inside):
//   row image   [chunk = col / 8][row ^ kswz(chunk)][8]  : A operand with the rows as MFMA rows (16-byte reads, conflict-free)
//   col image   [ks][g][pcol][8], element e <-> row 32 ks + 16 (e / 4) + 4 g + e % 4, pcol = 16 (col % 4) + (col % 64) / 4
//               (+ 64-blocks): A operand with the COLUMNS as MFMA rows (reduction over the tile's rows); the column
//               permutation makes the transposing 8-byte stores of 16 neighbouring lanes hit 16 different 16-byte slots.
#include <algorithm>
#include "common.h"

namespace {

#ifndef SATT_FLASH_PROBE
#define SATT_FLASH_PROBE 0
#endif
constexpr int FHD = 128, FT = 64, FNT = 256;
constexpr float FLOG2E = 1.4426950408889634f;
typedef __attribute__((ext_vector_type(4))) unsigned int fu32x4_t;

__device__ __forceinline__ int kswz(int c) { return ((c & 3) << 1) ^ ((c >> 2) & 1); }

// Staging is split into a LOAD phase (global -> registers, branch-free: rows clamped to the last valid one, masked later) and a
// STORE phase (registers -> bf16 LDS image).  A load inside `if (row < nvalid)` is waited for at the end of its branch: the
// fused form cost one L2 round trip per group - 24 serial trips per tile in the dK/dV kernel, which made the kernels latency
// bound (91 us for 5 GFLOP).  The split also lets a kernel request the NEXT tile before it computes on the current one.
// Workgroup barriers in the tile loops are lds_barrier() (s_waitcnt lgkmcnt(0); s_barrier): they order the LDS images only.
// __syncthreads() also drains vmcnt, i.e. it waited for the NEXT tile's loads issued just in front of it - the prefetch never
// overlapped anything and every 64-row iteration paid a full memory round trip (r3: 4.7 -> see DESIGN 9.5 us per iteration).
struct RowRegs { float4 v0[4], v1[4]; };      // rows r0 .. r0+63 of a matrix with row stride ld (floats)
struct ColRegs { float4 v[2][4]; };
__device__ __forceinline__ void load_rows(const float* __restrict__ src, int64_t ld, int nvalid, int tid, RowRegs& R) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int e = tid + FNT * g, row = min(e >> 4, nvalid - 1), chunk = e & 15;
    const float4* p = reinterpret_cast<const float4*>(src + (int64_t)row * ld + chunk * 8);
    R.v0[g] = p[0]; R.v1[g] = p[1];
  }
}
__device__ __forceinline__ void store_rows(const RowRegs& R, int nvalid, uint16_t* __restrict__ dst, int tid) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int e = tid + FNT * g, row = e >> 4, chunk = e & 15;
    fu32x4_t w;
    w[0] = pack_bf16x2(R.v0[g].x, R.v0[g].y); w[1] = pack_bf16x2(R.v0[g].z, R.v0[g].w);
    w[2] = pack_bf16x2(R.v1[g].x, R.v1[g].y); w[3] = pack_bf16x2(R.v1[g].z, R.v1[g].w);
    if (row >= nvalid) w = (fu32x4_t){0u, 0u, 0u, 0u};        // rows >= nvalid read as zero
    *reinterpret_cast<fu32x4_t*>(dst + (chunk * FT + (row ^ kswz(chunk & 7))) * 8) = w;
  }
}
__device__ __forceinline__ void load_cols(const float* __restrict__ src, int64_t ld, int nvalid, int tid, ColRegs& R) {
  const int cq = tid & 31;                       // columns 4 cq .. 4 cq + 3
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rq = (tid >> 5) + 8 * pass;        // rows 4 rq .. 4 rq + 3
#pragma unroll
    for (int r = 0; r < 4; ++r)
      R.v[pass][r] = *reinterpret_cast<const float4*>(src + (int64_t)min(4 * rq + r, nvalid - 1) * ld + 4 * cq);
  }
}
__device__ __forceinline__ void store_cols(const ColRegs& R, int nvalid, uint16_t* __restrict__ dst, int tid) {
  const int cq = tid & 31;
  const int pc0 = (cq >> 4) * 64 + (cq & 15);    // physical column of (4 cq + ii): pc0 + 16 ii
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rq = (tid >> 5) + 8 * pass;
    const int ks = rq >> 3, q8 = rq & 7, eh = q8 >> 2, g = q8 & 3;
    float f[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = 4 * rq + r < nvalid;
      const float4 v = R.v[pass][r];
      f[4 * r] = ok ? v.x : 0.f; f[4 * r + 1] = ok ? v.y : 0.f; f[4 * r + 2] = ok ? v.z : 0.f; f[4 * r + 3] = ok ? v.w : 0.f;
    }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      uint2 w;
      w.x = pack_bf16x2(f[ii], f[4 + ii]); w.y = pack_bf16x2(f[8 + ii], f[12 + ii]);
      *reinterpret_cast<uint2*>(dst + (((ks * 4 + g) * FHD + pc0 + 16 * ii) * 8) + 4 * eh) = w;
    }
  }
}
// A fragment of the row image: rows 16 t + (lane & 15), columns 32 ks + 8 g .. + 8
__device__ __forceinline__ bf16x8_t frag_rows(const uint16_t* img, int t, int ks, int lane) {
  const int chunk = 4 * ks + (lane >> 4), row = 16 * t + (lane & 15);
  return *reinterpret_cast<const bf16x8_t*>(img + (chunk * FT + (row ^ kswz(chunk & 7))) * 8);
}
// A fragment of the column image: physical columns 16 n + (lane & 15), rows of step ks in the permuted enumeration
__device__ __forceinline__ bf16x8_t frag_cols(const uint16_t* img, int n, int ks, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(img + ((ks * 4 + (lane >> 4)) * FHD + 16 * n + (lane & 15)) * 8);
}
// B fragment from global memory: column = row `row` of the matrix (valid or zero), elements 32 ks + 8 g .. + 8 of it
__device__ __forceinline__ bf16x8_t frag_global(const float* __restrict__ rowp, bool valid, int ks, int lane) {
  // (rowp is always a readable row: callers pass a clamped row pointer and the validity separately)
  const float4* p = reinterpret_cast<const float4*>(rowp + 32 * ks + 8 * (lane >> 4));
  const float4 v0 = p[0], v1 = p[1];
  fu32x4_t w;
  w[0] = pack_bf16x2(v0.x, v0.y); w[1] = pack_bf16x2(v0.z, v0.w);
  w[2] = pack_bf16x2(v1.x, v1.y); w[3] = pack_bf16x2(v1.z, v1.w);
  if (!valid) w = (fu32x4_t){0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8_t, w);
}
__device__ __forceinline__ bf16x8_t pack_b(const f32x4_t& lo, const f32x4_t& hi) {
  fu32x4_t w;
  w[0] = pack_bf16x2(lo[0], lo[1]); w[1] = pack_bf16x2(lo[2], lo[3]);
  w[2] = pack_bf16x2(hi[0], hi[1]); w[3] = pack_bf16x2(hi[2], hi[3]);
  return __builtin_bit_cast(bf16x8_t, w);
}
// reductions over the 4 lane groups of a column (lanes l, l ^ 16, l ^ 32, l ^ 48)
__device__ __forceinline__ float col_max(float v, int lane) {
  v = fmaxf(v, swz_xor(v, 16));
  return fmaxf(v, __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(v))));
}
__device__ __forceinline__ float col_sum(float v, int lane) {
  v += swz_xor(v, 16);
  return v + __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(v)));
}
// logical column of physical row pr of a column image (inverse of the staging permutation)
__device__ __forceinline__ int unperm(int pr) { return (pr & ~63) | ((pr & 15) << 2) | ((pr & 63) >> 4); }

// ---- the same staging for operands that already ARE bf16 in memory (the backward's K | V | Q and d o copies: the forward
// kernel writes the first while it stages them, the delta pass the second): half the bytes per tile, no conversion pass, half
// the prefetch registers.  Bit-identical results: the fp32 path rounds to nearest-even at the same point.
struct RowRegsB { fu32x4_t v[4]; };
struct ColRegsB { uint2 v[2][4]; };
__device__ __forceinline__ void load_rows(const uint16_t* __restrict__ src, int64_t ld, int nvalid, int tid, RowRegsB& R) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int e = tid + FNT * g, row = min(e >> 4, nvalid - 1), chunk = e & 15;
    R.v[g] = *reinterpret_cast<const fu32x4_t*>(src + (int64_t)row * ld + chunk * 8);
  }
}
__device__ __forceinline__ void store_rows(const RowRegsB& R, int nvalid, uint16_t* __restrict__ dst, int tid) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int e = tid + FNT * g, row = e >> 4, chunk = e & 15;
    fu32x4_t w = R.v[g];
    if (row >= nvalid) w = (fu32x4_t){0u, 0u, 0u, 0u};
    *reinterpret_cast<fu32x4_t*>(dst + (chunk * FT + (row ^ kswz(chunk & 7))) * 8) = w;
  }
}
__device__ __forceinline__ void load_cols(const uint16_t* __restrict__ src, int64_t ld, int nvalid, int tid, ColRegsB& R) {
  const int cq = tid & 31;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rq = (tid >> 5) + 8 * pass;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      R.v[pass][r] = *reinterpret_cast<const uint2*>(src + (int64_t)min(4 * rq + r, nvalid - 1) * ld + 4 * cq);
  }
}
__device__ __forceinline__ void store_cols(const ColRegsB& R, int nvalid, uint16_t* __restrict__ dst, int tid) {
  const int cq = tid & 31;
  const int pc0 = (cq >> 4) * 64 + (cq & 15);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rq = (tid >> 5) + 8 * pass;
    const int ks = rq >> 3, q8 = rq & 7, eh = q8 >> 2, g = q8 & 3;
    uint32_t lo[4], hi[4];                       // row r: columns (0, 1) and (2, 3) as packed pairs; rows >= nvalid read as zero
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = 4 * rq + r < nvalid;
      lo[r] = ok ? R.v[pass][r].x : 0u; hi[r] = ok ? R.v[pass][r].y : 0u;
    }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const uint32_t* w4 = (ii >> 1) ? hi : lo;
      uint2 w;
      if (ii & 1) { w.x = (w4[0] >> 16) | (w4[1] & 0xFFFF0000u); w.y = (w4[2] >> 16) | (w4[3] & 0xFFFF0000u); }
      else { w.x = (w4[0] & 0xFFFFu) | (w4[1] << 16); w.y = (w4[2] & 0xFFFFu) | (w4[3] << 16); }
      *reinterpret_cast<uint2*>(dst + (((ks * 4 + g) * FHD + pc0 + 16 * ii) * 8) + 4 * eh) = w;
    }
  }
}
__device__ __forceinline__ bf16x8_t frag_global(const uint16_t* __restrict__ rowp, bool valid, int ks, int lane) {
  fu32x4_t w = *reinterpret_cast<const fu32x4_t*>(rowp + 32 * ks + 8 * (lane >> 4));
  if (!valid) w = (fu32x4_t){0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8_t, w);
}
template <bool BF> struct FSrc { typedef float T; typedef RowRegs Row; typedef ColRegs Col; };
template <> struct FSrc<true> { typedef uint16_t T; typedef RowRegsB Row; typedef ColRegsB Col; };

struct FlashArgs {
  const float* k; const float* v; const float* q; int64_t ld;      // [B*T, .] rows, head h at column h * 128
  float* o; int64_t ldo; float* lse;                                // lse [B*H, T]: log2-domain log-sum-exp of the scaled scores
  const float* dout; const float* delta;                            // backward: d o [B*T, .] (stride ldo), delta [B*H, T]
  float* dk; float* dv; float* dq; int64_t ldd;
  int T, H, B; float scale; int causal;
  const uint16_t* kb; const uint16_t* vb; const uint16_t* qb; int64_t ldb;     // bf16 copies of k, v, q (backward source; forward: written if kvqb_out)
  const uint16_t* doutb; int64_t ldob;                                        // bf16 copy of d o (written by the delta pass, read by the backward)
  uint16_t* kb_out; uint16_t* vb_out; uint16_t* qb_out;                       // forward: bf16 copies to write (stride ldb), or null
  int tile_lo, tile_n;          // backward: the launch covers key / query tiles [tile_lo, tile_lo + tile_n) (causal suffix-first split)
  uint32_t thresh; float dscale; uint32_t stream; const uint32_t* seed;
};

// 1-D grid of nt * BH workgroups -> (tile, batch-head).  Workgroup i runs on XCD i % 8 and every XCD has its own L2: with
// the natural (tile fastest) order the tiles of one (batch, head) land on different XCDs and each of them pulls that
// head's K / V (or Q / dO) from HBM again - rocprofv3 FETCH_SIZE showed 141-154 MB per launch against 53-79 MB of
// algorithmic bytes.  Here all tiles of a (batch, head) share an XCD (BH % 8 == 0; plain order otherwise).
__device__ __forceinline__ void flash_block(int nt, int BH, int& tile, int& bh, int lin = (int)blockIdx.x) {
  if ((BH & 7) == 0) { const int slot = lin >> 3; bh = (slot / nt) * 8 + (lin & 7); tile = slot - (slot / nt) * nt; }
  else { bh = lin / nt; tile = lin - bh * nt; }
}

// ------------------------------------------------------------------------------------------------ forward
// DROP (compile time): dropout on the probabilities.  The element loops below are BRANCH-FREE: with a run-time `if (a.thresh)` /
// `thresh == 0 || keep(...)` per element the compiler emitted two or three branches around every hash - ~200 taken branches per
// 64-key iteration of the backward bodies, which is where a third of their time went (ISA read of r3's last session).
template <bool DROP>
__attribute__((amdgpu_waves_per_eu(2, 2)))     // 2 workgroups per CU (the bf16-copy stores must not cost the second wave per SIMD)
__global__ __launch_bounds__(FNT) void flash_fwd_k(const FlashArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * FT * FHD];
  uint16_t* Ks = lds;                 // row image of the K tile
  uint16_t* Vt = lds + FT * FHD;      // column image of the V tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int T = a.T, nqt = (T + FT - 1) / FT;
  int bx, bh; flash_block(nqt, a.B * a.H, bx, bh);
  const int b = bh / a.H, h = bh - b * a.H;
  const int qt = nqt - 1 - bx, i0 = qt * FT;   // longest (latest) query tiles first
  const float* K = a.k + (int64_t)b * T * a.ld + h * FHD;
  const float* V = a.v + (int64_t)b * T * a.ld + h * FHD;
  const float* Q = a.q + (int64_t)b * T * a.ld + h * FHD;
  const int iq = i0 + 16 * wave + (lane & 15);
  bf16x8_t qb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qb[ks] = frag_global(Q + (int64_t)min(iq, T - 1) * a.ld, iq < T, ks, lane);
  if (a.qb_out && iq < T) {        // bf16 copy of this workgroup's query rows (the backward reads the copies)
    uint16_t* qo = a.qb_out + ((int64_t)b * T + iq) * a.ldb + h * FHD;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) *reinterpret_cast<bf16x8_t*>(qo + 32 * ks + 8 * g) = qb[ks];
  }
  const uint32_t seed = (a.thresh && a.seed) ? *a.seed : 0u;
  const float c2 = a.scale * FLOG2E;
  float m = -INFINITY, lsum = 0.f;
  f32x4_t oacc[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) oacc[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nkt = a.causal ? qt + 1 : nqt;
  RowRegs rk; ColRegs rv;
  load_rows(K, a.ld, T, tid, rk);
  load_cols(V, a.ld, T, tid, rv);
  for (int kt = 0; kt < nkt; ++kt) {
    const int j0 = kt * FT;
    lds_barrier();
    store_rows(rk, T - j0, Ks, tid);
    store_cols(rv, T - j0, Vt, tid);
    if (a.kb_out && kt == qt) {       // the diagonal tile: this workgroup writes the bf16 copies of its K and V rows
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int e = tid + FNT * gq, row = e >> 4, chunk = e & 15;
        if (j0 + row < T) {
          fu32x4_t w;
          w[0] = pack_bf16x2(rk.v0[gq].x, rk.v0[gq].y); w[1] = pack_bf16x2(rk.v0[gq].z, rk.v0[gq].w);
          w[2] = pack_bf16x2(rk.v1[gq].x, rk.v1[gq].y); w[3] = pack_bf16x2(rk.v1[gq].z, rk.v1[gq].w);
          *reinterpret_cast<fu32x4_t*>(a.kb_out + ((int64_t)b * T + j0 + row) * a.ldb + h * FHD + chunk * 8) = w;
        }
      }
      const int cq = tid & 31;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * ((tid >> 5) + 8 * pass) + r;
          if (j0 + row < T) {
            const float4 v = rv.v[pass][r];
            uint2 w; w.x = pack_bf16x2(v.x, v.y); w.y = pack_bf16x2(v.z, v.w);
            *reinterpret_cast<uint2*>(a.vb_out + ((int64_t)b * T + j0 + row) * a.ldb + h * FHD + 4 * cq) = w;
          }
        }
    }
    if (kt + 1 < nkt) {               // the next tile travels while this one is computed
      load_rows(K + (int64_t)(j0 + FT) * a.ld, a.ld, T - j0 - FT, tid, rk);
      load_cols(V + (int64_t)(j0 + FT) * a.ld, a.ld, T - j0 - FT, tid, rv);
    }
    lds_barrier();
    f32x4_t s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, j, ks, lane), qb[ks], s[j], 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j0 + 16 * j + 4 * g + r;
        const bool ok = (key < T) & (!a.causal | (key <= iq));
        s[j][r] = ok ? s[j][r] * c2 : -INFINITY;
        mx = fmaxf(mx, s[j][r]);
      }
    mx = col_max(mx, lane);
    const float mn = fmaxf(m, mx);                 // finite from the first tile on: key 0 is valid for every query
    const float alpha = exp2f_(m - mn);
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = exp2f_(s[j][r] - mn);     // masked entries: exp2(-inf) = 0
        rs += pv;
        float pd = pv;
        if constexpr (DROP) {
          const int key = j0 + 16 * j + 4 * g + r;
          pd = satt_keep(seed, a.stream, (uint32_t)(((int64_t)bh * T + iq) * T + key), a.thresh) ? pv * a.dscale : 0.f;
        }
        s[j][r] = pd;
      }
    rs = col_sum(rs, lane);
    lsum = lsum * alpha + rs;
    m = mn;
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[n][r] *= alpha;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t pb = pack_b(s[2 * ks], s[2 * ks + 1]);
#pragma unroll
      for (int n = 0; n < 8; ++n) oacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Vt, n, ks, lane), pb, oacc[n], 0, 0, 0);
    }
  }
  if (iq < T) {
    const float inv = 1.f / lsum;
    float* orow = a.o + ((int64_t)b * T + iq) * a.ldo + h * FHD;
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // physical rows 64 bq + 16 nn + 4 g + r (nn = 0..3) are the logical columns 64 bq + 16 g + 4 r + nn
        const float4 v = make_float4(oacc[4 * bq][r] * inv, oacc[4 * bq + 1][r] * inv, oacc[4 * bq + 2][r] * inv, oacc[4 * bq + 3][r] * inv);
        *reinterpret_cast<float4*>(orow + 64 * bq + 16 * g + 4 * r) = v;
      }
    if (g == 0) a.lse[(int64_t)bh * T + iq] = m + log2f(lsum);
  }
}

// ------------------------------------------------------------------------------------------------ delta = rowsum(dO * O)
// rows [r0, r1) of every sample only (a tile-range launch needs the sums - and the bf16 d o - of ITS query rows: the suffix launch of
// the split decoder head, which the recurrent pipeline waits for, then pays for 64 rows per sample instead of all of them)
__global__ __launch_bounds__(256) void flash_delta_k(const float* __restrict__ o, const float* __restrict__ dout, int64_t ldo,
                                                     float* __restrict__ delta, int B, int T, int H,
                                                     uint16_t* __restrict__ doutb, int64_t ldob, int r0, int r1) {
  const int lane = threadIdx.x & 63;
  const int64_t w = blockIdx.x * 4 + (threadIdx.x >> 6);        // (b * (r1 - r0) + i - r0) * H + h
  const int nr = r1 - r0;
  if (w >= (int64_t)B * nr * H) return;
  const int64_t rr = w / H; const int h = (int)(w - rr * H);
  const int64_t row = (rr / nr) * T + r0 + (rr % nr);
  const float2 x = *reinterpret_cast<const float2*>(o + row * ldo + h * FHD + 2 * lane);
  const float2 y = *reinterpret_cast<const float2*>(dout + row * ldo + h * FHD + 2 * lane);
  if (doutb) *reinterpret_cast<uint32_t*>(doutb + row * ldob + h * FHD + 2 * lane) = pack_bf16x2(y.x, y.y);    // bf16 copy of d o
  const float s = wave_sum(x.x * y.x + x.y * y.y);
  if (lane == 0) { const int64_t b = row / T; delta[(b * H + h) * T + (row - b * T)] = s; }
}

// ------------------------------------------------------------------------------------------------ dK, dV
constexpr int DKV_LDS = (4 * FT * FHD) * 2 + 2 * FT * 4;          // Qs, Qt, Ds, Dt (bf16) + lse, delta of the query tile
template <bool BF, bool DROP>
__device__ __forceinline__ void flash_dkv_body(const FlashArgs& a, uint16_t* dyn, int lin) {
  typedef typename FSrc<BF>::T ST;
  const int64_t sld = BF ? a.ldb : a.ld, dld = BF ? a.ldob : a.ldo;
  uint16_t* Qs = dyn; uint16_t* Qt = dyn + FT * FHD; uint16_t* Ds = dyn + 2 * FT * FHD; uint16_t* Dt = dyn + 3 * FT * FHD;
  float* Ls = reinterpret_cast<float*>(dyn + 4 * FT * FHD); float* dl = Ls + FT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int T = a.T, nt = (T + FT - 1) / FT;
  int kt, bh; flash_block(a.tile_n, a.B * a.H, kt, bh, lin);
  kt += a.tile_lo;
  const int b = bh / a.H, h = bh - b * a.H;
  const int j0 = kt * FT;      // causal: early key tiles (most work) first
  const ST* K = (BF ? (const ST*)a.kb : (const ST*)a.k) + (int64_t)b * T * sld + h * FHD;
  const ST* V = (BF ? (const ST*)a.vb : (const ST*)a.v) + (int64_t)b * T * sld + h * FHD;
  const ST* Q = (BF ? (const ST*)a.qb : (const ST*)a.q) + (int64_t)b * T * sld + h * FHD;
  const ST* DO = (BF ? (const ST*)a.doutb : (const ST*)a.dout) + (int64_t)b * T * dld + h * FHD;
  const int key = j0 + 16 * wave + (lane & 15);
  bf16x8_t kb[4], vb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kb[ks] = frag_global(K + (int64_t)min(key, T - 1) * sld, key < T, ks, lane);
    vb[ks] = frag_global(V + (int64_t)min(key, T - 1) * sld, key < T, ks, lane);
  }
  const uint32_t seed = (a.thresh && a.seed) ? *a.seed : 0u;
  const float c2 = a.scale * FLOG2E;
  f32x4_t dvt[8], dkt[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) { dvt[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dkt[n] = dvt[n]; }
  // Register stages of the query tiles in flight: with bf16 sources a stage is 66 registers, so TWO tiles travel while one is
  // computed (the loads of tile t + 2 are issued as soon as tile t has been written to LDS: a whole iteration more to arrive)
  struct Stage { typename FSrc<BF>::Row rq, rd; typename FSrc<BF>::Col cq, cd; float rl, rdl; };
  constexpr int NS = BF ? 2 : 1;
  Stage st[NS];
  auto request = [&](Stage& S, int i0) {        // every global operand of the query tile at i0 (branch-free, clamped)
    load_rows(Q + (int64_t)i0 * sld, sld, T - i0, tid, S.rq);
    load_cols(Q + (int64_t)i0 * sld, sld, T - i0, tid, S.cq);
    load_rows(DO + (int64_t)i0 * dld, dld, T - i0, tid, S.rd);
    load_cols(DO + (int64_t)i0 * dld, dld, T - i0, tid, S.cd);
    const int64_t li = (int64_t)bh * T + min(i0 + (tid & (FT - 1)), T - 1);
    S.rl = a.lse[li]; S.rdl = a.delta[li];
  };
  const int qt0 = a.causal ? kt : 0;
  auto step = [&](Stage& S, int qt) {
    const int i0 = qt * FT;
    lds_barrier();
    store_rows(S.rq, T - i0, Qs, tid);
    store_cols(S.cq, T - i0, Qt, tid);
    store_rows(S.rd, T - i0, Ds, tid);
    store_cols(S.cd, T - i0, Dt, tid);
    if (tid < FT) {
      const bool ok = i0 + tid < T;
      Ls[tid] = ok ? S.rl : 0.f;
      dl[tid] = ok ? S.rdl : 0.f;
    }
    if (qt + NS < nt) request(S, i0 + NS * FT);        // the tile NS ahead travels while this one (and the next) is computed
    lds_barrier();
    f32x4_t pd[4], ds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Qs, i, ks, lane), kb[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ds, i, ks, lane), vb[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 16 * i + 4 * g + r, q = i0 + ql;
        const bool ok = (q < T) & (key < T) & (!a.causal | (key <= q));
        const float ev = exp2f_(s[r] * c2 - Ls[ql]);          // (masked entries may overflow to inf: selected away, never multiplied)
        const float pv = ok ? ev : 0.f;
        if constexpr (DROP) {
          const bool keep = satt_keep(seed, a.stream, (uint32_t)(((int64_t)bh * T + q) * T + key), a.thresh);
          pd[i][r] = keep ? pv * a.dscale : 0.f;
          ds[i][r] = pv * ((keep ? dp[r] * a.dscale : 0.f) - dl[ql]);
        } else {
          pd[i][r] = pv;
          ds[i][r] = pv * (dp[r] - dl[ql]);
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t pb = pack_b(pd[2 * ks], pd[2 * ks + 1]), sb = pack_b(ds[2 * ks], ds[2 * ks + 1]);
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        dvt[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Dt, n, ks, lane), pb, dvt[n], 0, 0, 0);
        dkt[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Qt, n, ks, lane), sb, dkt[n], 0, 0, 0);
      }
    }
  };
  request(st[0], qt0 * FT);
  if (NS == 2 && qt0 + 1 < nt) request(st[NS - 1], (qt0 + 1) * FT);
  for (int qt = qt0; qt < nt; qt += NS) {
    step(st[0], qt);
    if (NS == 2 && qt + 1 < nt) step(st[NS - 1], qt + 1);
  }
  if (key < T) {
    float* dkr = a.dk + ((int64_t)b * T + key) * a.ldd + h * FHD;
    float* dvr = a.dv + ((int64_t)b * T + key) * a.ldd + h * FHD;
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 64 * bq + 16 * g + 4 * r;
        *reinterpret_cast<float4*>(dvr + col) = make_float4(dvt[4 * bq][r], dvt[4 * bq + 1][r], dvt[4 * bq + 2][r], dvt[4 * bq + 3][r]);
        *reinterpret_cast<float4*>(dkr + col) = make_float4(dkt[4 * bq][r] * a.scale, dkt[4 * bq + 1][r] * a.scale,
                                                             dkt[4 * bq + 2][r] * a.scale, dkt[4 * bq + 3][r] * a.scale);
      }
  }
}

// ------------------------------------------------------------------------------------------------ dQ
template <bool BF, bool DROP>
__device__ __forceinline__ void flash_dq_body(const FlashArgs& a, uint16_t* lds, int lin) {
  typedef typename FSrc<BF>::T ST;
  const int64_t sld = BF ? a.ldb : a.ld, dld = BF ? a.ldob : a.ldo;
  uint16_t* Ks = lds; uint16_t* Kt = lds + FT * FHD; uint16_t* Vs = lds + 2 * FT * FHD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int T = a.T;
  int bx, bh; flash_block(a.tile_n, a.B * a.H, bx, bh, lin);
  const int b = bh / a.H, h = bh - b * a.H;
  const int qt = a.tile_lo + a.tile_n - 1 - bx, i0 = qt * FT;
  const ST* K = (BF ? (const ST*)a.kb : (const ST*)a.k) + (int64_t)b * T * sld + h * FHD;
  const ST* V = (BF ? (const ST*)a.vb : (const ST*)a.v) + (int64_t)b * T * sld + h * FHD;
  const ST* Q = (BF ? (const ST*)a.qb : (const ST*)a.q) + (int64_t)b * T * sld + h * FHD;
  const ST* DO = (BF ? (const ST*)a.doutb : (const ST*)a.dout) + (int64_t)b * T * dld + h * FHD;
  const int iq = i0 + 16 * wave + (lane & 15);
  bf16x8_t qb[4], dob[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qb[ks] = frag_global(Q + (int64_t)min(iq, T - 1) * sld, iq < T, ks, lane);
    dob[ks] = frag_global(DO + (int64_t)min(iq, T - 1) * dld, iq < T, ks, lane);
  }
  const float Lq_ = a.lse[(int64_t)bh * T + min(iq, T - 1)], dqr_ = a.delta[(int64_t)bh * T + min(iq, T - 1)];
  const float Lq = iq < T ? Lq_ : 0.f, dq_ = iq < T ? dqr_ : 0.f;
  const uint32_t seed = (a.thresh && a.seed) ? *a.seed : 0u;
  const float c2 = a.scale * FLOG2E;
  f32x4_t dqt[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) dqt[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nkt = a.causal ? qt + 1 : (T + FT - 1) / FT;
  struct Stage { typename FSrc<BF>::Row rk, rv; typename FSrc<BF>::Col ck; };
  constexpr int NS = BF ? 2 : 1;                // key tiles in flight (see flash_dkv_body)
  Stage st[NS];
  auto request = [&](Stage& S, int j0) {
    load_rows(K + (int64_t)j0 * sld, sld, T - j0, tid, S.rk);
    load_cols(K + (int64_t)j0 * sld, sld, T - j0, tid, S.ck);
    load_rows(V + (int64_t)j0 * sld, sld, T - j0, tid, S.rv);
  };
  auto step = [&](Stage& S, int kt) {
    const int j0 = kt * FT;
    lds_barrier();
#if SATT_FLASH_PROBE == 2          // timing probes (wrong results): 2 = stage the first tile only, 3 = no loads inside the loop
    if (kt == 0) {
#endif
    store_rows(S.rk, T - j0, Ks, tid);
    store_cols(S.ck, T - j0, Kt, tid);
    store_rows(S.rv, T - j0, Vs, tid);
#if SATT_FLASH_PROBE == 2
    }
#endif
#if SATT_FLASH_PROBE != 3
    if (kt + NS < nkt) request(S, j0 + NS * FT);
#endif
    lds_barrier();
    f32x4_t ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, j, ks, lane), qb[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Vs, j, ks, lane), dob[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j0 + 16 * j + 4 * g + r;
        const bool ok = (iq < T) & (key < T) & (!a.causal | (key <= iq));
#if SATT_FLASH_PROBE == 1          // 1 = no exp2 / mask / dropout arithmetic
        ds[j][r] = s[r] * dp[r];
#else
        const float ev = exp2f_(s[r] * c2 - Lq);
        const float pv = ok ? ev : 0.f;
        float dpr = dp[r];
        if constexpr (DROP)
          dpr = satt_keep(seed, a.stream, (uint32_t)(((int64_t)bh * T + iq) * T + key), a.thresh) ? dpr * a.dscale : 0.f;
        ds[j][r] = pv * (dpr - dq_);
#endif
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t sb = pack_b(ds[2 * ks], ds[2 * ks + 1]);
#pragma unroll
      for (int n = 0; n < 8; ++n) dqt[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Kt, n, ks, lane), sb, dqt[n], 0, 0, 0);
    }
  };
  request(st[0], 0);
  if (NS == 2 && 1 < nkt) request(st[NS - 1], FT);
  for (int kt = 0; kt < nkt; kt += NS) {
    step(st[0], kt);
    if (NS == 2 && kt + 1 < nkt) step(st[NS - 1], kt + 1);
  }
  if (iq < T) {
    float* dqr = a.dq + ((int64_t)b * T + iq) * a.ldd + h * FHD;
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(dqr + 64 * bq + 16 * g + 4 * r) =
            make_float4(dqt[4 * bq][r] * a.scale, dqt[4 * bq + 1][r] * a.scale, dqt[4 * bq + 2][r] * a.scale, dqt[4 * bq + 3][r] * a.scale);
  }
}

inline bool fl16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// dK / dV tiles and dQ tiles in ONE launch (r3): the two passes are independent given delta, and the dK/dV pass alone leaves
// most CUs idle behind its longest workgroups (key tile 0 walks every query tile).  Workgroups [0, n) are key tiles (longest
// first), [n, 2n) query tiles; n is a multiple of 8 whenever the XCD placement of flash_block applies, so lin & 7 keeps its meaning.
template <bool BF, bool DROP>
#ifdef SATT_FLASH_WAVES
__attribute__((amdgpu_waves_per_eu(SATT_FLASH_WAVES, SATT_FLASH_WAVES)))       // occupancy experiment (tools/build_variant.sh)
#endif
__global__ __launch_bounds__(FNT) void flash_bwd_k(const FlashArgs a, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) uint16_t dyn[];
  const int lin = (int)blockIdx.x;
  if (lin < ntiles) flash_dkv_body<BF, DROP>(a, dyn, lin);
  else flash_dq_body<BF, DROP>(a, dyn, lin - ntiles);
}

// kvqb: optional bf16 copies of k | v | q written by the forward kernel (same head layout, row stride ldb elements)
static int flash_fwd_launch(const float* k, const float* v, const float* q, int64_t ld, float* o, int64_t ldo, float* lse,
                            int B, int T, int H, int head_dim, float scale, int causal, uint32_t drop_thresh,
                            float drop_scale, uint32_t drop_stream, const uint32_t* seed, uint16_t* kb, uint16_t* vb,
                            uint16_t* qb, int64_t ldb, void* stream) {
  if (!k || !v || !q || !o || !lse || B <= 0 || T <= 0 || H <= 0) return SATT_E_BADARG;
  if (head_dim != FHD) return SATT_E_UNSUPPORTED;
  if (ld % 4 || ldo % 4 || !fl16(k) || !fl16(v) || !fl16(q) || !fl16(o)) return SATT_E_UNSUPPORTED;
  if ((int64_t)B * H > 65535 || (int64_t)B * H * T * T >= (1ll << 32)) return SATT_E_UNSUPPORTED;   // dropout counter is 32 bits
  if (kb && (!vb || !qb || ldb % 8 || !fl16(kb) || !fl16(vb) || !fl16(qb))) return SATT_E_BADARG;
  FlashArgs a{};
  a.k = k; a.v = v; a.q = q; a.ld = ld; a.o = o; a.ldo = ldo; a.lse = lse; a.T = T; a.H = H; a.B = B; a.scale = scale; a.causal = causal;
  a.thresh = drop_thresh; a.dscale = drop_scale; a.stream = drop_stream; a.seed = seed;
  a.kb_out = kb; a.vb_out = vb; a.qb_out = qb; a.ldb = ldb;
  if (drop_thresh) hipLaunchKernelGGL(flash_fwd_k<true>, dim3(((T + FT - 1) / FT) * B * H), dim3(FNT), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(flash_fwd_k<false>, dim3(((T + FT - 1) / FT) * B * H), dim3(FNT), 0, (hipStream_t)stream, a);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_flash_attn_fwd(const float* k, const float* v, const float* q, int64_t ld, float* o, int64_t ldo, float* lse,
                                   int B, int T, int H, int head_dim, float scale, int causal, uint32_t drop_thresh,
                                   float drop_scale, uint32_t drop_stream, const uint32_t* seed, void* stream) {
  return flash_fwd_launch(k, v, q, ld, o, ldo, lse, B, T, H, head_dim, scale, causal, drop_thresh, drop_scale, drop_stream, seed,
                          nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int satt_flash_attn_fwd_b(const float* k, const float* v, const float* q, int64_t ld, float* o, int64_t ldo, float* lse,
                                     int B, int T, int H, int head_dim, float scale, int causal, uint32_t drop_thresh,
                                     float drop_scale, uint32_t drop_stream, const uint32_t* seed, uint16_t* kb, uint16_t* vb,
                                     uint16_t* qb, int64_t ldb, void* stream) {
  if (!kb) return SATT_E_BADARG;
  return flash_fwd_launch(k, v, q, ld, o, ldo, lse, B, T, H, head_dim, scale, causal, drop_thresh, drop_scale, drop_stream, seed,
                          kb, vb, qb, ldb, stream);
}

// kb / vb / qb (bf16 copies from satt_flash_attn_fwd_b, stride ldb) and doutb (scratch, stride ldob: written by the delta pass of
// the launch that has with_delta set, read by this and every later launch of the same (o, dout) pair): the bf16-source backward
static int flash_bwd_launch(const float* k, const float* v, const float* q, int64_t ld, const float* o, const float* dout,
                            int64_t ldo, const float* lse, float* delta, float* dk, float* dv, float* dq, int64_t ldd,
                            int B, int T, int H, int head_dim, float scale, int causal, uint32_t drop_thresh,
                            float drop_scale, uint32_t drop_stream, const uint32_t* seed, int tile_lo, int tile_hi,
                            int with_delta, const uint16_t* kb, const uint16_t* vb, const uint16_t* qb, int64_t ldb,
                            uint16_t* doutb, int64_t ldob, void* stream) {
  const bool bf = kb != nullptr;
  if ((!bf && (!k || !v || !q)) || !o || !dout || !lse || !delta || !dk || !dv || !dq || B <= 0 || T <= 0 || H <= 0) return SATT_E_BADARG;
  if (head_dim != FHD) return SATT_E_UNSUPPORTED;
  if (ldo % 4 || ldd % 4 || !fl16(o) || !fl16(dout) || !fl16(dk) || !fl16(dv) || !fl16(dq)) return SATT_E_UNSUPPORTED;
  if (!bf && (ld % 4 || !fl16(k) || !fl16(v) || !fl16(q))) return SATT_E_UNSUPPORTED;
  if (bf && (!vb || !qb || !doutb || ldb % 8 || ldob % 8 || !fl16(kb) || !fl16(vb) || !fl16(qb) || !fl16(doutb))) return SATT_E_BADARG;
  if ((int64_t)B * H > 65535 || (int64_t)B * H * T * T >= (1ll << 32)) return SATT_E_UNSUPPORTED;
  const int nt = (T + FT - 1) / FT;
  if (tile_lo < 0 || tile_hi > nt || tile_lo >= tile_hi) return SATT_E_BADARG;
  // a proper sub-range is only closed under the causal mask: key tile j takes query tiles >= j, query tile i key tiles <= i,
  // so the rows of the tiles of ANY range are final after a launch over that range
  if (!causal && (tile_lo != 0 || tile_hi != nt)) return SATT_E_BADARG;
  FlashArgs a{};
  a.k = k; a.v = v; a.q = q; a.ld = ld; a.ldo = ldo; a.lse = const_cast<float*>(lse); a.dout = dout; a.delta = delta;
  a.dk = dk; a.dv = dv; a.dq = dq; a.ldd = ldd; a.T = T; a.H = H; a.B = B; a.scale = scale; a.causal = causal;
  a.tile_lo = tile_lo; a.tile_n = tile_hi - tile_lo;
  a.thresh = drop_thresh; a.dscale = drop_scale; a.stream = drop_stream; a.seed = seed;
  a.kb = kb; a.vb = vb; a.qb = qb; a.ldb = ldb; a.doutb = doutb; a.ldob = ldob;
  hipStream_t s = (hipStream_t)stream;
  // row sums of o * d o [+ the bf16 d o] for the query rows of THIS tile range.  Under the causal mask a range reads the rows of
  // its own tiles and of LATER tiles (key tile j takes query tiles >= j): launches must therefore run from the last range to the
  // first, each with with_delta set - what the split decoder head does; the full range covers every row in one launch.
  const int r0 = tile_lo * FT, r1 = std::min(tile_hi * FT, T);
  const int64_t nw = (int64_t)B * (r1 - r0) * H;
  if (with_delta)
    hipLaunchKernelGGL(flash_delta_k, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, s, o, dout, ldo, delta, B, T, H,
                       bf ? doutb : (uint16_t*)nullptr, ldob, r0, r1);
  static_assert(DKV_LDS >= 3 * FT * FHD * 2, "the dQ body fits the dK/dV body's LDS");
  const int ntiles = a.tile_n * B * H;
#define SATT_FLASH_BWD(BFV, DRV)                                                                                        \
  do {                                                                                                                    \
    (void)hipFuncSetAttribute((const void*)flash_bwd_k<BFV, DRV>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);   \
    hipLaunchKernelGGL((flash_bwd_k<BFV, DRV>), dim3(2 * ntiles), dim3(FNT), DKV_LDS, s, a, ntiles);                      \
  } while (0)
  if (bf) { if (drop_thresh) SATT_FLASH_BWD(true, true); else SATT_FLASH_BWD(true, false); }
  else { if (drop_thresh) SATT_FLASH_BWD(false, true); else SATT_FLASH_BWD(false, false); }
#undef SATT_FLASH_BWD
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_flash_attn_bwd_tiles(const float* k, const float* v, const float* q, int64_t ld, const float* o, const float* dout,
                                         int64_t ldo, const float* lse, float* delta, float* dk, float* dv, float* dq, int64_t ldd,
                                         int B, int T, int H, int head_dim, float scale, int causal, uint32_t drop_thresh,
                                         float drop_scale, uint32_t drop_stream, const uint32_t* seed, int tile_lo, int tile_hi,
                                         int with_delta, void* stream) {
  return flash_bwd_launch(k, v, q, ld, o, dout, ldo, lse, delta, dk, dv, dq, ldd, B, T, H, head_dim, scale, causal, drop_thresh,
                          drop_scale, drop_stream, seed, tile_lo, tile_hi, with_delta, nullptr, nullptr, nullptr, 0, nullptr, 0, stream);
}

extern "C" int satt_flash_attn_bwd_tiles_b(const uint16_t* kb, const uint16_t* vb, const uint16_t* qb, int64_t ldb, const float* o,
                                           const float* dout, int64_t ldo, uint16_t* doutb, int64_t ldob, const float* lse,
                                           float* delta, float* dk, float* dv, float* dq, int64_t ldd, int B, int T, int H,
                                           int head_dim, float scale, int causal, uint32_t drop_thresh, float drop_scale,
                                           uint32_t drop_stream, const uint32_t* seed, int tile_lo, int tile_hi, int with_delta,
                                           void* stream) {
  if (!kb) return SATT_E_BADARG;
  return flash_bwd_launch(nullptr, nullptr, nullptr, 0, o, dout, ldo, lse, delta, dk, dv, dq, ldd, B, T, H, head_dim, scale, causal,
                          drop_thresh, drop_scale, drop_stream, seed, tile_lo, tile_hi, with_delta, kb, vb, qb, ldb, doutb, ldob, stream);
}

extern "C" int satt_flash_attn_bwd(const float* k, const float* v, const float* q, int64_t ld, const float* o, const float* dout,
                                   int64_t ldo, const float* lse, float* delta, float* dk, float* dv, float* dq, int64_t ldd,
                                   int B, int T, int H, int head_dim, float scale, int causal, uint32_t drop_thresh,
                                   float drop_scale, uint32_t drop_stream, const uint32_t* seed, void* stream) {
  return satt_flash_attn_bwd_tiles(k, v, q, ld, o, dout, ldo, lse, delta, dk, dv, dq, ldd, B, T, H, head_dim, scale, causal,
                                   drop_thresh, drop_scale, drop_stream, seed, 0, T > 0 ? (T + FT - 1) / FT : 0, 1, stream);
}
