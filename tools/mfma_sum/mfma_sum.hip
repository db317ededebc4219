// mfma_sum.hip -- how does v_mfma_f32_32x32x16_f16 sum its 16 products and the accumulator input?  Probes the internal
// alignment width: products p_k = a_k * b_k (exact in any format) of very different magnitudes that cancel, a small
// survivor, and an accumulator input.  Prints the result next to the exact sum.  (round 6: is it safe to let large terms
// cancel INSIDE one matrix instruction?)    hipcc --offload-arch=gfx950 -O2 -o mfma_sum mfma_sum.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_one(const _Float16 *a16, const _Float16 *b16, float c_in, float *out) {
  const int lane = threadIdx.x;
  const int kh = lane >> 5;   // k = 8 kh + i
  f16x8 a, b;
  for (int i = 0; i < 8; i++) {
    a[i] = a16[8 * kh + i];
    b[i] = b16[8 * kh + i];
  }
  f32x16 c;
  for (int i = 0; i < 16; i++) c[i] = c_in;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];
}

static float run(const std::vector<double> &av, const std::vector<double> &bv, float c_in) {
  _Float16 ha[16], hb[16];
  for (int i = 0; i < 16; i++) {
    ha[i] = (_Float16)(i < (int)av.size() ? av[i] : 0.0);
    hb[i] = (_Float16)(i < (int)bv.size() ? bv[i] : 0.0);
  }
  _Float16 *da, *db;
  float *dout, h = 0;
  hipMalloc(&da, 32); hipMalloc(&db, 32); hipMalloc(&dout, 4);
  hipMemcpy(da, ha, 32, hipMemcpyHostToDevice);
  hipMemcpy(db, hb, 32, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, da, db, c_in, dout);
  hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost);
  hipFree(da); hipFree(db); hipFree(dout);
  return h;
}

int main() {
  // 1. [+B, -B, t]: the survivor t = 2^-e against cancelling products of magnitude B = 2^9
  printf("case 1: products [+512, -512, 2^-e] (k = 0, 1, 2), c = 0: result / exact\n");
  for (int e = 4; e <= 28; e += 2) {
    const double t = std::ldexp(1.0, -e);
    // t = 2^-7 * 2^-(e-7) keeps both factors normal fp16 numbers up to e = 21
    float r = run({512.0, -512.0, std::ldexp(1.0, -7)}, {1.0, 1.0, std::ldexp(1.0, -(e - 7))}, 0.0f);
    printf("  e %2d  got %.10g  exact %.10g  %s\n", e, r, t, r == (float)t ? "exact" : "LOST/ROUNDED");
  }
  printf("case 2: the same with the survivor FIRST (k = 0) and the pair at k = 8, 9 (the other half of the lanes)\n");
  for (int e = 10; e <= 24; e += 2) {
    std::vector<double> a(16, 0.0), b(16, 0.0);
    a[0] = std::ldexp(1.0, -7); b[0] = std::ldexp(1.0, -(e - 7));
    a[8] = 512.0; b[8] = 1.0; a[9] = -512.0; b[9] = 1.0;
    float r = run(a, b, 0.0f);
    printf("  e %2d  got %.10g  exact %.10g\n", e, r, std::ldexp(1.0, -e));
  }
  printf("case 3: products [+512, -512], c = 2^-e\n");
  for (int e = 10; e <= 30; e += 4) {
    float r = run({512.0, -512.0}, {1.0, 1.0}, (float)std::ldexp(1.0, -e));
    printf("  e %2d  got %.10g  exact %.10g\n", e, r, std::ldexp(1.0, -e));
  }
  printf("case 4: products [+512 (1 + 2^-10), -512, 1 + 2^-10 + ...]: mantissa bits kept of a sum of magnitude ~1 next to 512s\n");
  {
    // p0 = 512 + 0.5, p1 = -512, p2 = 2^-e: exact = 0.5 + 2^-e
    for (int e = 12; e <= 26; e += 2) {
      float r = run({512.5, -512.0, std::ldexp(1.0, -7)}, {1.0, 1.0, std::ldexp(1.0, -(e - 7))}, 0.0f);
      printf("  e %2d  got %.10g  exact %.10g\n", e, r, 0.5 + std::ldexp(1.0, -e));
    }
  }
  printf("case 5: 16 products alternating +-B (1 + k 2^-10) with B = 256: exact sum vs result\n");
  {
    std::vector<double> a(16), b(16, 1.0);
    double exact = 0;
    for (int k = 0; k < 16; k++) {
      a[k] = (k & 1 ? -1.0 : 1.0) * 256.0 * (1.0 + k * std::ldexp(1.0, -10));
      exact += (double)(_Float16)a[k];
    }
    float r = run(a, b, 0.3f);
    printf("  got %.10g  exact %.10g (+0.3 = %.10g)\n", r, exact, exact + (double)0.3f);
  }
  printf("case 6: c = 300.3 (large accumulator), products [+2^-e]: does a tiny product survive next to a large c? (f32 ulp at 300 = 3e-5)\n");
  for (int e = 12; e <= 20; e += 2) {
    float r = run({std::ldexp(1.0, -7)}, {std::ldexp(1.0, -(e - 7))}, 300.3f);
    printf("  e %2d  got %.10g  exact %.10g\n", e, r, (double)300.3f + std::ldexp(1.0, -e));
  }
  printf("case 7: c = -300, products [+512, -212 + 2^-e ...] -> result near 2^-e: is c part of the same wide sum?\n");
  for (int e = 8; e <= 24; e += 4) {
    // p0 = 512, p1 = -212, c = -300 -> 0; p2 = 2^-e
    float r = run({512.0, -212.0, std::ldexp(1.0, -7)}, {1.0, 1.0, std::ldexp(1.0, -(e - 7))}, -300.0f);
    printf("  e %2d  got %.10g  exact %.10g\n", e, r, std::ldexp(1.0, -e));
  }
  printf("case 8-11: rounding mode and guard bits (max term 512: window lsb 2^-14 = 6.1035e-05)\n");
  {
    const double t3 = 3.0 * std::ldexp(1.0, -9), s7 = std::ldexp(1.0, -7), h = std::ldexp(1.0, -8);
    float r;
    r = run({512.0, -512.0, t3}, {1.0, 1.0, s7}, 0.0f);
    printf("  [512,-512, 0.75 lsb]              got %.10g (trunc 0, round 6.1035e-05, exact 4.5776e-05)\n", r);
    r = run({512.0, -512.0, t3, t3}, {1.0, 1.0, s7, s7}, 0.0f);
    printf("  [512,-512, 0.75 lsb x2]           got %.10g (each trunc 0, each round 1.2207e-04, exact 9.1553e-05)\n", r);
    r = run({512.0, -512.0, -t3}, {1.0, 1.0, s7}, 0.0f);
    printf("  [512,-512, -0.75 lsb]             got %.10g\n", r);
    r = run({512.0, -512.0, h, h}, {1.0, 1.0, s7, s7}, 0.0f);
    printf("  [512,-512, 0.5 lsb x2]            got %.10g (exact 6.1035e-05)\n", r);
    r = run({512.0, -512.0, h, h, h, h}, {1.0, 1.0, s7, s7, s7, s7}, 0.0f);
    printf("  [512,-512, 0.5 lsb x4]            got %.10g (exact 1.2207e-04)\n", r);
    r = run({512.0, -512.0, h / 2, h / 2, h / 2, h / 2}, {1.0, 1.0, s7, s7, s7, s7}, 0.0f);
    printf("  [512,-512, 0.25 lsb x4]           got %.10g (exact 6.1035e-05)\n", r);
    r = run({512.0, -512.0, h / 4, h / 4, h / 4, h / 4, h / 4, h / 4, h / 4, h / 4}, {1.0, 1.0, s7, s7, s7, s7, s7, s7, s7, s7}, 0.0f);
    printf("  [512,-512, 0.125 lsb x8]          got %.10g (exact 6.1035e-05)\n", r);
    r = run({512.0, t3}, {1.0, s7}, 0.0f);
    printf("  [512, 0.75 lsb] (no cancellation) got %.10g (512 + ? : f32 ulp at 512 = 6.1035e-05)\n", r);
    r = run({512.0, -512.0}, {1.0, 1.0}, (float)(0.75 * std::ldexp(1.0, -14)));
    printf("  [512,-512] + c = 0.75 lsb         got %.10g\n", r);
    r = run({512.0, -512.0, t3}, {1.0, 1.0, s7}, (float)(0.75 * std::ldexp(1.0, -14)));
    printf("  [512,-512, 0.75 lsb] + c 0.75 lsb got %.10g (exact 9.1553e-05)\n", r);
    r = run({256.0, 256.0, -512.0, t3}, {1.0, 1.0, 1.0, s7}, 0.0f);
    printf("  [256,256,-512, 0.75 lsb of 2^9]   got %.10g (max term 512? window by max product)\n", r);
    r = run({256.0, 256.0, 256.0, -768.0 + 0, t3}, {1.0, 1.0, 1.0, 1.0, s7}, 0.0f);
    printf("  [256 x3, -768, 3*2^-16]           got %.10g (max term 768 -> e 9)\n", r);
    r = run({256.0, -256.0, t3}, {1.0, 1.0, s7}, 0.0f);
    printf("  [256,-256, 3*2^-16 = 1.5 lsb(2^-15)] got %.10g (exact 4.5776e-05, trunc 3.0518e-05)\n", r);
  }
  return 0;
}
