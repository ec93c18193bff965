// Thin GEMMs of the flow's 3-/12-channel edges: one dimension of the contraction is <= 128 (27 = 9 taps x 3 channels,
// 108 = 9 x 12), so the launch is bound by streaming the wide [pixels, 512] tensor through HBM, not by the MFMA pipe.
// igemm.hip tries these kernels first (indm_thin_gemm returns THIN_NA when the problem is not one of theirs).
#pragma once
#include "indm_common.h"

struct ThinGemmArgs {
  const float* A;
  const float* G;  // optional gate on A (same layout)
  const float* B;
  float* C;
  int M, N, K;
  int transA, transB;  // 0: [M,K] / [K,N] row-major; 1: stored [K,M] / [N,K]
  long lda, ldb, ldc;
  int batch;
  float alpha, beta;
  const float* bias;
  const float* bias2;
  int rpg;
  const float* R;
  long ldr;
  float* ws;  // split-K workspace (weight-grad forms)
  long ws_floats;
  // implicit im2col (X != NULL): A[p, tap*cC + c] = X[n, h - csign*(kh-1), w - csign*(kw-1), c] of an NHWC image with
  // cC channels (3x3 taps, zero padding), p = (n, h, w); A / lda are ignored, K = 9*cC, M = images*cH*cW
  const float* X;
  int cH, cW, cC, csign;
  // "bf16 compute / fp32 accumulate" mode: both MFMA operands are rounded to bf16 (RNE) at the fragment read and
  // multiplied on the exact fp32 MFMA -- products of bf16 values are exact in fp32, so the result equals a bf16 MFMA's.
  // (These kernels are HBM-bound either way.)
  int bf16;
  // im2col convolution only: non-NULL -> the Sin activation as the epilogue: C receives a = sin(2 pi v) / (2 pi), act_d receives
  // d = cos(2 pi v) (both [M, ldc]), v = the convolution + biases, which is not written (alpha must be 1, no residual)
  float* act_d;
  // experiment knobs of the im2col kernel (INDM_THIN_TUNE, tools/kflat_probe.py): bits 0-7 start delay of every second
  // workgroup in units of 64 cycles x 16, bit 8 non-temporal stores, bit 9 uncounted waits, bits 10-11 workgroups per CU - 1
  int tune;
};

enum { THIN_NA = 1 };

// *variant names the kernel that ran ("thin_k128x128" | "thin_wgrad32x128" | "thin_wgrad128x128")
int indm_thin_gemm(const ThinGemmArgs& a, hipStream_t st, const char** variant);
// 3x3 / pad 1 / stride 1 convolution whose INPUT is skinny (C <= 3, or C <= 12 with C % 4 == 0), as a thin-K GEMM with
// the im2col gather inside the operand loader (THIN_NA when the geometry is not covered)
int indm_thin_conv3x3(const ThinGemmArgs& a, hipStream_t st);
