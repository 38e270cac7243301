// The highway stack of the CBHG encoder in ONE launch per direction (reference modules/module.py:87-90, HighwayNet :258-277):
//   y = relu(x Wh + bh) * t + x * (1 - t),  t = sigmoid(x Wt + bt),   4 layers of 128 units, rows = B * Ti.
// Every layer is row-wise, so a workgroup carries its 32 rows through ALL layers: the layer input stays in registers in the
// MFMA result layout (the gate needs x at exactly the positions of the product's output), only its bf16 image goes through LDS
// as the A operand.  The unfused form was 2 launches per layer and direction (GEMM + gate kernel, 8 + 8 per step, each a few
// microseconds of work behind a launch boundary); the pre-activations z and the layer outputs are still written - the weight
// gradients (side stream GEMMs) and the backward pass read them.
//
// v_mfma_f32_16x16x32_bf16 operand layout (as in flash.hip): A lane l = row (l & 15), k = 8 (l >> 4) .. + 8; B lane l = column
// (l & 15), same k; C lane l = rows 4 (l >> 4) + r, column (l & 15).  Wave w owns the unit columns 32 w .. 32 w + 31 (two tiles)
// of both halves of W, so hp and tp of one unit meet in one lane.
//   forward : z[32 x 256] = x[32 x 128] W[128 x 256]      B operand = rows of the transposed bf16 shadow  Wt [256][128]
//   backward: dx[32 x 128] = g (1 - t) + dz[32 x 256] W^T  B operand = rows of the plain bf16 shadow       Wn [128][256]
// bf16 operands, fp32 accumulation: the arithmetic of the GEMM kernels in bf16 mode (the engine keeps the unfused form in
// exact-fp32 mode).
#include "common.h"

namespace {

constexpr int HW_H = 128, HW_RT = 32, HW_NT = 256;
struct HwArgs {
  satt_highway_layer L[SATT_HIGHWAY_MAX_LAYERS];
  const float* x0; const float* dy; float* dx;
  int nl, rows;
};

__device__ __forceinline__ float hw_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }     // (the gate kernels' form)

__global__ __launch_bounds__(HW_NT) void highway_stack_fwd_k(const HwArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t xs[HW_RT][HW_H + 8];      // bf16 image of the layer input (A operand)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cl = lane & 15;
  const int r0 = blockIdx.x * HW_RT, rows = a.rows;
  float x[2][2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(r0 + 16 * rt + 4 * g + r, rows - 1), col = 16 * (2 * wave + j) + cl;
        x[rt][j][r] = a.x0[(size_t)row * HW_H + col];
      }
  bf16x8_t bh[4][2], bt[4][2], nh[4][2], nt[4][2];
  auto load_w = [&](const uint16_t* Wt, bf16x8_t (&h)[4][2], bf16x8_t (&t)[4][2]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = 16 * (2 * wave + j) + cl;
        h[ks][j] = *reinterpret_cast<const bf16x8_t*>(Wt + (size_t)n * HW_H + 32 * ks + 8 * g);
        t[ks][j] = *reinterpret_cast<const bf16x8_t*>(Wt + (size_t)(HW_H + n) * HW_H + 32 * ks + 8 * g);
      }
  };
  load_w(reinterpret_cast<const uint16_t*>(a.L[0].Wt), bh, bt);
#pragma unroll 1
  for (int n = 0; n < a.nl; ++n) {
    const satt_highway_layer& L = a.L[n];
    float bias_h[2], bias_t[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int col = 16 * (2 * wave + j) + cl; bias_h[j] = L.b[col]; bias_t[j] = L.b[HW_H + col]; }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[16 * rt + 4 * g + r][16 * (2 * wave + j) + cl] = f2bf(x[rt][j][r]);
    // the next layer's weight tiles travel while this layer is computed (lds_barrier does not wait for global loads)
    if (n + 1 < a.nl) load_w(reinterpret_cast<const uint16_t*>(a.L[n + 1].Wt), nh, nt);
    lds_barrier();
    f32x4_t ah[2][2], at[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int j = 0; j < 2; ++j) { ah[rt][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; at[rt][j] = ah[rt][j]; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t av[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) av[rt] = *reinterpret_cast<const bf16x8_t*>(&xs[16 * rt + cl][32 * ks + 8 * g]);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          ah[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[rt], bh[ks][j], ah[rt][j], 0, 0, 0);
          at[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[rt], bt[ks][j], at[rt][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = r0 + 16 * rt + 4 * g + r, col = 16 * (2 * wave + j) + cl;
          const float hp = ah[rt][j][r] + bias_h[j], tp = at[rt][j][r] + bias_t[j];
          const float t = hw_sigmoid(tp);
          const float y = fmaxf(hp, 0.f) * t + x[rt][j][r] * (1.f - t);
          if (row < rows) {
            L.z[(size_t)row * 2 * HW_H + col] = hp; L.z[(size_t)row * 2 * HW_H + HW_H + col] = tp;
            L.y[(size_t)row * HW_H + col] = y;
          }
          x[rt][j][r] = y;
        }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) { bh[ks][j] = nh[ks][j]; bt[ks][j] = nt[ks][j]; }
    lds_barrier();      // every wave has read this layer's image before the next one is written
  }
}

__global__ __launch_bounds__(HW_NT) void highway_stack_bwd_k(const HwArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t ds[HW_RT][2 * HW_H + 8];  // bf16 image of d z (A operand)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, cl = lane & 15;
  const int r0 = blockIdx.x * HW_RT, rows = a.rows;
  float gr[2][2][4], hp[2][2][4], tp[2][2][4], xi[2][2][4];
  auto load_act = [&](int n) {       // pre-activations and input of layer n at this thread's positions (rows clamped)
    const satt_highway_layer& L = a.L[n];
    const float* xin = n == 0 ? a.x0 : a.L[n - 1].y;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = min(r0 + 16 * rt + 4 * g + r, rows - 1), col = 16 * (2 * wave + j) + cl;
          hp[rt][j][r] = L.z[(size_t)row * 2 * HW_H + col]; tp[rt][j][r] = L.z[(size_t)row * 2 * HW_H + HW_H + col];
          xi[rt][j][r] = xin[(size_t)row * HW_H + col];
        }
  };
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(r0 + 16 * rt + 4 * g + r, rows - 1), col = 16 * (2 * wave + j) + cl;
        gr[rt][j][r] = a.dy[(size_t)row * HW_H + col];
      }
  load_act(a.nl - 1);
#pragma unroll 1
  for (int n = a.nl - 1; n >= 0; --n) {
    const satt_highway_layer& L = a.L[n];
    // B tiles of this layer: W[x column 16 cj + cl][z column 32 ks + 8 g .. + 8], 8 K steps x 2 column tiles
    bf16x8_t bw[8][2];
    const uint16_t* Wn = reinterpret_cast<const uint16_t*>(L.Wn);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bw[ks][j] = *reinterpret_cast<const bf16x8_t*>(Wn + (size_t)(16 * (2 * wave + j) + cl) * 2 * HW_H + 32 * ks + 8 * g);
    f32x4_t acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = r0 + 16 * rt + 4 * g + r, col = 16 * (2 * wave + j) + cl, rl = 16 * rt + 4 * g + r;
          const float t = hw_sigmoid(tp[rt][j][r]), h = fmaxf(hp[rt][j][r], 0.f), gg = gr[rt][j][r];
          const float dzh = hp[rt][j][r] > 0.f ? gg * t : 0.f;
          const float dzt = gg * (h - xi[rt][j][r]) * t * (1.f - t);
          if (row < rows) { L.dz[(size_t)row * 2 * HW_H + col] = dzh; L.dz[(size_t)row * 2 * HW_H + HW_H + col] = dzt; }
          ds[rl][col] = f2bf(dzh); ds[rl][HW_H + col] = f2bf(dzt);
          acc[rt][j][r] = gg * (1.f - t);
        }
    if (n > 0) load_act(n - 1);        // the next layer's operands travel during the product
    lds_barrier();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bf16x8_t av[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) av[rt] = *reinterpret_cast<const bf16x8_t*>(&ds[16 * rt + cl][32 * ks + 8 * g]);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[rt], bw[ks][j], acc[rt][j], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) gr[rt][j][r] = acc[rt][j][r];
    lds_barrier();
  }
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + 16 * rt + 4 * g + r, col = 16 * (2 * wave + j) + cl;
        if (row < rows) a.dx[(size_t)row * HW_H + col] = gr[rt][j][r];
      }
}

inline int hw_check(const satt_highway_layer* L, int nl, int rows, int H, bool bwd) {
  if (!L || nl <= 0 || rows <= 0) return SATT_E_BADARG;
  if (H != HW_H || nl > SATT_HIGHWAY_MAX_LAYERS) return SATT_E_UNSUPPORTED;
  for (int n = 0; n < nl; ++n) {
    if (!L[n].z || !L[n].y || (bwd ? (!L[n].Wn || !L[n].dz) : (!L[n].Wt || !L[n].b))) return SATT_E_BADARG;
    if (((uintptr_t)L[n].Wt | (uintptr_t)L[n].Wn) & 15) return SATT_E_UNSUPPORTED;
  }
  return SATT_OK;
}

}  // namespace

extern "C" int satt_highway_stack_fwd(const float* x, const satt_highway_layer* layers, int nlayers, int rows, int H, void* stream) {
  int rc = hw_check(layers, nlayers, rows, H, false);
  if (rc) return rc;
  if (!x) return SATT_E_BADARG;
  HwArgs a{};
  for (int n = 0; n < nlayers; ++n) a.L[n] = layers[n];
  a.x0 = x; a.nl = nlayers; a.rows = rows;
  hipLaunchKernelGGL(highway_stack_fwd_k, dim3((rows + HW_RT - 1) / HW_RT), dim3(HW_NT), 0, (hipStream_t)stream, a);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_highway_stack_bwd(const float* dy, const float* x, const satt_highway_layer* layers, int nlayers, int rows, int H,
                                      float* dx, void* stream) {
  int rc = hw_check(layers, nlayers, rows, H, true);
  if (rc) return rc;
  if (!dy || !x || !dx) return SATT_E_BADARG;
  HwArgs a{};
  for (int n = 0; n < nlayers; ++n) a.L[n] = layers[n];
  a.x0 = x; a.dy = dy; a.dx = dx; a.nl = nlayers; a.rows = rows;
  hipLaunchKernelGGL(highway_stack_bwd_k, dim3((rows + HW_RT - 1) / HW_RT), dim3(HW_NT), 0, (hipStream_t)stream, a);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
