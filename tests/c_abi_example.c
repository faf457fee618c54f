/* A plain-C caller of libmultike_hip.so — no Python, no torch: the drop-in boundary really is a C ABI.
 * Allocates device buffers with the HIP runtime, runs one relation-view step (fused triple step + row updates) on a
 * tiny problem and checks the loss against the closed form for this input.  Built and run by
 * tests/test_c_abi_gpu.py:   hipcc tests/c_abi_example.c -Iinclude -Lmultike_amd -lmultike_hip -o /tmp/c_abi_example */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "multike_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define MK(x) do { int r_ = (x); if (r_ != 0) { printf("mke error %d: %s (line %d)\n", r_, mke_last_error(), __LINE__); return 3; } } while (0)

int main(void) {
  const int dim = 75, stride = 80, n_ent = 64, n_rel = 4, P = 3, N = 2;
  float* h_ent = (float*)calloc((size_t)n_ent * stride, 4);
  float* h_rel = (float*)calloc((size_t)n_rel * stride, 4);
  /* entity e = unit vector along axis e; relation r = unit vector along axis 70+r */
  for (int e = 0; e < n_ent; ++e) h_ent[e * stride + e] = 2.5f;   /* raw norm 2.5: read through l2_normalize -> 1 */
  for (int r = 0; r < n_rel; ++r) h_rel[r * stride + 70 + r] = 0.5f;
  int32_t ph[] = {1, 2, 3}, pr[] = {0, 1, 2}, pt[] = {4, 5, 3};       /* third positive has h == t */
  int32_t nh[] = {1, 9, 2, 2, 8, 3}, nr[] = {0, 0, 1, 1, 2, 2}, nt[] = {7, 4, 6, 5, 3, 3};  /* 4th negative == its positive */
  float *ent, *rel, *acc_e, *acc_r, *g_e, *g_r;
  int32_t *te, *tr, *d_idx;
  double* lp;
  size_t be = (size_t)n_ent * stride * 4, br = (size_t)n_rel * stride * 4;
  CK(hipMalloc(&ent, be)); CK(hipMalloc(&rel, br)); CK(hipMalloc(&acc_e, be)); CK(hipMalloc(&acc_r, br));
  CK(hipMalloc(&g_e, be)); CK(hipMalloc(&g_r, br)); CK(hipMalloc(&te, n_ent * 4)); CK(hipMalloc(&tr, n_rel * 4));
  CK(hipMalloc(&lp, MKE_LOSS_PARTIALS * 8)); CK(hipMalloc(&d_idx, 64 * 4));
  CK(hipMemcpy(ent, h_ent, be, hipMemcpyHostToDevice)); CK(hipMemcpy(rel, h_rel, br, hipMemcpyHostToDevice));
  CK(hipMemset(g_e, 0, be)); CK(hipMemset(g_r, 0, br)); CK(hipMemset(te, 0, n_ent * 4)); CK(hipMemset(tr, 0, n_rel * 4));
  float* h_acc = (float*)malloc(be);
  for (size_t i = 0; i < (size_t)n_ent * stride; ++i) h_acc[i] = 0.1f;
  CK(hipMemcpy(acc_e, h_acc, be, hipMemcpyHostToDevice)); CK(hipMemcpy(acc_r, h_acc, br, hipMemcpyHostToDevice));
  int32_t h_idx[64];
  memcpy(h_idx, ph, 12); memcpy(h_idx + 3, pr, 12); memcpy(h_idx + 6, pt, 12);
  memcpy(h_idx + 9, nh, 24); memcpy(h_idx + 15, nr, 24); memcpy(h_idx + 21, nt, 24);
  CK(hipMemcpy(d_idx, h_idx, sizeof(h_idx), hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  if (mke_version() != MKE_VERSION) { printf("version mismatch\n"); return 4; }
  MK(mke_triple_score_fwd_bwd(ent, n_ent, 1, rel, n_rel, 1, stride, dim, d_idx, d_idx + 3, d_idx + 6, NULL, P, d_idx + 9,
                              d_idx + 15, d_idx + 21, NULL, (int64_t)P * N, N, 1.0f, g_e, g_r, 1, te, tr, 1, lp, st));
  MK(mke_rows_update(ent, acc_e, g_e, 1, te, 1, n_ent, stride, dim, 1, MKE_OPT_ADAGRAD, 0.01f, st));
  MK(mke_rows_update(rel, acc_r, g_r, 1, tr, 1, n_rel, stride, dim, 1, MKE_OPT_ADAGRAD, 0.01f, st));
  CK(hipStreamSynchronize(st));
  double h_lp[MKE_LOSS_PARTIALS], loss = 0;
  CK(hipMemcpy(h_lp, lp, sizeof(h_lp), hipMemcpyDeviceToHost));
  for (int i = 0; i < MKE_LOSS_PARTIALS; ++i) loss += h_lp[i];
  /* orthonormal rows: ||h + r - t||^2 = 3 when h != t, 1 when h == t */
  const double sp3 = log(1 + exp(3.0)), sp1 = log(1 + exp(1.0)), sn3 = log(1 + exp(-3.0)), sn1 = log(1 + exp(-1.0));
  /* positives: (1,0,4)->3, (2,1,5)->3, (3,2,3)->1 ; negatives: (1,0,7) 3, (9,0,4) 3, (2,1,6) 3, (2,1,5) 3, (8,2,3) 3, (3,2,3) 1 */
  const double expect = 2 * sp3 + sp1 + 5 * sn3 + sn1;
  printf("loss %.9f expected %.9f\n", loss, expect);
  if (fabs(loss - expect) > 2e-6 * expect) { printf("MISMATCH\n"); return 1; }
  /* an argument error comes back as a code + message, nothing is launched */
  if (mke_rows_update(ent, acc_e, g_e, 1, te, 2, n_ent, 75, dim, 1, MKE_OPT_ADAGRAD, 0.01f, st) != MKE_E_SHAPE) return 5;
  float* h_g = (float*)malloc(be);
  CK(hipMemcpy(h_g, g_e, be, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < (size_t)n_ent * stride; ++i) if (h_g[i] != 0.f) { printf("gradient scratch not consumed\n"); return 6; }
  printf("C ABI example ok\n");
  return 0;
}
