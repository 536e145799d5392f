/* Pure-C client of the C ABI (no Python, no torch): what a maintainer's FFI binding would do.
 *   sf_c_client <in.bin> <out.bin>
 * in.bin : int32 header {kind,D,C,H,T,K,NB,B,S,P,seed}, then float32 theta_mean[D] theta_std[D] x_mean[C] x_std[C],
 *          int32 perms[T*D], float32 flat[P], theta[B*D], x[B*C], lo[D], hi[D]
 * out.bin: float32 log_prob[B], then float32 samples[B*S*D] from sf_flow_sample (prior box lo..hi), int32 n_drawn[B]
 * Built with gcc and run by tests/test_gpu_c_client.py (libamdhip64 only supplies the device buffers). */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "synference_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, sf_last_error()); return 2; } } while (0)
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

static void* rd(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(4); }
  return p;
}

int main(int argc, char** argv) {
  if (argc != 3) return 1;
  FILE* fi = fopen(argv[1], "rb");
  if (!fi) return 1;
  int32_t* h = (int32_t*)rd(fi, 11 * sizeof(int32_t));
  const int kind = h[0], D = h[1], C = h[2], H = h[3], T = h[4], K = h[5], NB = h[6], B = h[7], S = h[8], P = h[9], seed = h[10];
  float* tm = (float*)rd(fi, D * 4); float* ts = (float*)rd(fi, D * 4);
  float* xm = (float*)rd(fi, C * 4); float* xs = (float*)rd(fi, C * 4);
  int32_t* perms = (int32_t*)rd(fi, (size_t)T * D * 4);
  float* flat = (float*)rd(fi, (size_t)P * 4);
  float* theta = (float*)rd(fi, (size_t)B * D * 4);
  float* x = (float*)rd(fi, (size_t)B * C * 4);
  float* lo = (float*)rd(fi, D * 4); float* hi = (float*)rd(fi, D * 4);
  fclose(fi);

  sf_flow_desc d = {0};
  d.kind = kind; d.D = D; d.C = C; d.H = H; d.T = T; d.K = K; d.NB = NB;
  d.tail_bound = 3.0f; d.min_bin_width = d.min_bin_height = d.min_derivative = 1e-3f; d.maf_eps = d.lu_eps = 1e-3f;
  d.theta_mean = tm; d.theta_std = ts; d.x_mean = xm; d.x_std = xs; d.perms = kind == SF_MAF ? perms : NULL;
  sf_flow* f = NULL;
  CHECK(sf_flow_create(&d, &f));
  if (sf_flow_num_params(f) != P) { fprintf(stderr, "P mismatch %lld\n", (long long)sf_flow_num_params(f)); return 5; }
  CHECK(sf_flow_set_params(f, flat, P, /*is_device=*/0, NULL));

  float *d_theta, *d_x, *d_lp, *d_lo, *d_hi, *d_s; int32_t* d_nd;
  HIPCHECK(hipMalloc((void**)&d_theta, (size_t)B * D * 4)); HIPCHECK(hipMalloc((void**)&d_x, (size_t)B * C * 4));
  HIPCHECK(hipMalloc((void**)&d_lp, (size_t)B * 4)); HIPCHECK(hipMalloc((void**)&d_lo, D * 4)); HIPCHECK(hipMalloc((void**)&d_hi, D * 4));
  HIPCHECK(hipMalloc((void**)&d_s, (size_t)B * S * D * 4)); HIPCHECK(hipMalloc((void**)&d_nd, (size_t)B * 4));
  HIPCHECK(hipMemcpy(d_theta, theta, (size_t)B * D * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d_x, x, (size_t)B * C * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d_lo, lo, D * 4, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(d_hi, hi, D * 4, hipMemcpyHostToDevice));

  CHECK(sf_flow_log_prob(f, d_theta, d_x, B, d_lp, NULL));
  int64_t unfilled = -1;
  CHECK(sf_flow_sample(f, d_x, B, S, d_lo, d_hi, (uint64_t)seed, 64, d_s, d_nd, &unfilled, NULL));
  HIPCHECK(hipDeviceSynchronize());

  float* lp = (float*)malloc((size_t)B * 4); float* s = (float*)malloc((size_t)B * S * D * 4); int32_t* nd = (int32_t*)malloc((size_t)B * 4);
  HIPCHECK(hipMemcpy(lp, d_lp, (size_t)B * 4, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(s, d_s, (size_t)B * S * D * 4, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(nd, d_nd, (size_t)B * 4, hipMemcpyDeviceToHost));
  FILE* fo = fopen(argv[2], "wb");
  fwrite(lp, 4, B, fo); fwrite(s, 4, (size_t)B * S * D, fo); fwrite(nd, 4, B, fo);
  fclose(fo);
  printf("%s unfilled=%lld\n", sf_version(), (long long)unfilled);
  sf_flow_destroy(f);
  return 0;
}
