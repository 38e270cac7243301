// Large-tile bf16 GEMM families (gemm_tile.hip) that satt_gemm dispatches to when a problem fits them.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/satt_hip.h"

// true: the problem was launched on stream s.  false: not eligible (nothing launched) - use the generic kernel.
bool satt_gemm_tile_rk(const satt_gemm_params& p, hipStream_t s);
bool satt_gemm_tile_dw(const satt_gemm_params& p, hipStream_t s);
// 0: generic kernel, 1: gemm_rk_k, 2: gemm_dw_k, 3: conv_bank_fwd_k (host-only; p must already carry satt_gemm's defaults)
int satt_gemm_tile_path(const satt_gemm_params& p);
// floats of workspace (satt_gemm_params.ws) a split reduction of this problem can use instead of atomics; 0: none
int64_t satt_gemm_tile_ws_floats(const satt_gemm_params& p);
