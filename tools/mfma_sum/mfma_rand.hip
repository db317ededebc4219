// mfma_rand.hip -- random cases through one v_mfma_f32_32x32x16_f16: inputs and result to gpurun_out/mfma_cases.bin
// (per case: 16 a, 16 b as float, c, result), for fitting the summation rule offline (tools/emu_mfma.py --fit).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_many(const _Float16 *a16, const _Float16 *b16, const float *c_in, float *out, int n) {
  const int lane = threadIdx.x;
  const int kh = lane >> 5;
  for (int t = blockIdx.x; t < n; t += gridDim.x) {
    f16x8 a, b;
    for (int i = 0; i < 8; i++) {
      a[i] = a16[t * 16 + 8 * kh + i];
      b[i] = b16[t * 16 + 8 * kh + i];
    }
    f32x16 c;
    for (int i = 0; i < 16; i++) c[i] = c_in[t];
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[t] = c[0];
  }
}

int main() {
  const int n = 20000;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> N(0.0, 1.0);
  std::vector<_Float16> ha(n * 16), hb(n * 16);
  std::vector<float> hc(n), ho(n), fa(n * 16), fb(n * 16);
  for (int t = 0; t < n; t++) {
    // pairs (lin, quad) that nearly cancel: lin = 2 m u, quad = -u^2 (scaled), magnitudes 2^-2 .. 2^9
    const double scale = std::ldexp(1.0, (int)(U(rng) * 12) - 2);
    for (int k = 0; k < 16; k += 2) {
      const double m = scale * (0.5 + U(rng)), z = N(rng) * std::sqrt(scale > 1 ? scale : 1.0) * 0.3;
      double a0 = m, b0 = m + z, a1 = -0.5, b1 = (m + z) * (m + z) * (1 + 0.01 * N(rng));
      if (std::fabs(b1) > 60000) b1 = 60000;
      ha[t * 16 + k] = (_Float16)a0; hb[t * 16 + k] = (_Float16)b0;
      ha[t * 16 + k + 1] = (_Float16)a1; hb[t * 16 + k + 1] = (_Float16)b1;
    }
    hc[t] = (float)(N(rng) * scale * 2.0);
    for (int k = 0; k < 16; k++) { fa[t * 16 + k] = (float)ha[t * 16 + k]; fb[t * 16 + k] = (float)hb[t * 16 + k]; }
  }
  _Float16 *da, *db; float *dc, *dout;
  hipMalloc(&da, n * 32); hipMalloc(&db, n * 32); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(da, ha.data(), n * 32, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), n * 32, hipMemcpyHostToDevice);
  hipMemcpy(dc, hc.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_many, dim3(256), dim3(64), 0, 0, da, db, dc, dout, n);
  hipMemcpy(ho.data(), dout, n * 4, hipMemcpyDeviceToHost);
  FILE *f = fopen("gpurun_out/mfma_cases.bin", "wb");
  if (!f) { perror("open"); return 1; }
  for (int t = 0; t < n; t++) {
    fwrite(&fa[t * 16], 4, 16, f); fwrite(&fb[t * 16], 4, 16, f); fwrite(&hc[t], 4, 1, f); fwrite(&ho[t], 4, 1, f);
  }
  fclose(f);
  printf("wrote %d cases\n", n);
  return 0;
}
