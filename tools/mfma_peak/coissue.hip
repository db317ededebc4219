// Co-issue probe for gfx950: what the matrix pipe sustains when the SAME wave interleaves vector work between its
// matrix instructions, as k_gmm_diag_score_pl does (per 30 v_mfma_f32_32x32x16_f16: 32 v_exp_f32 and ~64 plain VALU
// instructions).  Register operands only, no memory traffic, no barriers: whatever is lost here against the matrix-only
// stream is lost to instruction issue, not to the kernel's data movement.
//   per matrix instruction: E transcendentals (v_exp_f32) + A plain VALU instructions (v_add_f32), all independent of the
//   matrix results; CH accumulator chains per wave (2 in the scoring kernel); W waves per SIMD (2 in its 8-wave form).
//   hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip && ./coissue [ms_per_launch]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                      \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

template <int E, int A, int CH>
__global__ __launch_bounds__(512) void k_coissue(const unsigned *__restrict__ pat, float *__restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 a[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    a[k] = ((const u32x4 *)pat)[(t * 8 + k) & 16383];
    b[k] = ((const u32x4 *)pat)[(t * 8 + 4 + k) & 16383];
  }
  f32x16 c[4] = {{0}, {0}, {0}, {0}};
  float x[4] = {-1.0f, -2.0f, -3.0f, -0.5f}, s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const f16x8 av = __builtin_bit_cast(f16x8, a[r & 3]), bv = __builtin_bit_cast(f16x8, b[(r >> 2) & 3]);
      c[r % CH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c[r % CH], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < E; e++) asm volatile("v_exp_f32 %0, %1\n\ts_nop 0" : "=v"(x[(r + e) & 3]) : "v"(x[(r + e) & 3]));
#pragma unroll
      for (int e = 0; e < A; e++) asm volatile("v_add_f32 %0, %1, %2" : "=v"(s[(r + e) & 3]) : "v"(s[(r + e) & 3]), "v"(x[(r + e + 1) & 3]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float acc = s[0] + s[1] + s[2] + s[3] + x[0] + x[1] + x[2] + x[3];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc += c[k][e];
  out[t] = acc;
}

template <int E, int A, int CH>
static double run(int waves_per_simd, int cus, const unsigned *d_pat, float *d_out, double target_ms) {
  // workgroups of 8 waves: one per CU for 2 waves per SIMD, two per CU for 4; ONE wave per SIMD: workgroups of 4 waves
  const int threads = waves_per_simd == 1 ? 256 : 512;
  if (waves_per_simd > 1) cus = cus * waves_per_simd / 2;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  int iters = 500;
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_coissue<E, A, CH>), dim3(cus), dim3(threads), 0, 0, d_pat, d_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    iters = (int)(iters * target_ms / ms) + 1;
  }
  double sum = 0;
  const int reps = 6;
  for (int rep = 0; rep < reps; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_coissue<E, A, CH>), dim3(cus), dim3(threads), 0, 0, d_pat, d_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 1) sum += (double)cus * (threads / 64) * iters * 16.0 * 32768.0 / (ms * 1e-3) / 1e12;
  }
  return sum / (reps - 1);
}

int main(int argc, char **argv) {
  const double target_ms = argc > 1 ? atof(argv[1]) : 30.0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned *d_pat;
  float *d_out;
  CHECK(hipMalloc(&d_pat, 16384 * 16));
  CHECK(hipMalloc(&d_out, (size_t)cus * 2 * 512 * 4));
  std::vector<unsigned short> h(16384 * 8);
  srand(99);
  for (size_t i = 0; i < h.size(); i++) {   // fp16 values with random mantissas and exponents over 2^-12 .. 2^3
    const unsigned short mant = (unsigned short)(rand() & 0x3ff), ex = (unsigned short)(3 + rand() % 16), sg = (unsigned short)(rand() & 1);
    h[i] = (unsigned short)((sg << 15) | (ex << 10) | mant);
  }
  CHECK(hipMemcpy(d_pat, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  printf("%s, %d CUs: v_mfma_f32_32x32x16_f16 with vector work of the same wave between the matrix instructions\n", prop.name, cus);
  printf("%-58s %10s %10s\n", "per matrix instruction", "2 waves/SIMD", "4 waves/SIMD");
#define ROW(E, A, CH, label)                                                                       \
  {                                                                                                \
    const double t2 = run<E, A, CH>(2, cus, d_pat, d_out, target_ms);                              \
    const double t4 = run<E, A, CH>(4, cus, d_pat, d_out, target_ms);                              \
    printf("%-58s %7.0f TF %7.0f TF   (%.2f / %.2f of 2500)\n", label, t2, t4, t2 / 2500, t4 / 2500); \
  }
  ROW(0, 0, 4, "nothing, 4 accumulator chains")
  ROW(0, 0, 2, "nothing, 2 accumulator chains")
  ROW(0, 2, 2, "2 v_add_f32, 2 chains")
  ROW(1, 0, 2, "1 v_exp_f32, 2 chains")
  ROW(1, 2, 2, "1 v_exp_f32 + 2 v_add_f32, 2 chains (the scoring kernel's mix)")
  ROW(1, 3, 2, "1 v_exp_f32 + 3 v_add_f32, 2 chains (its masked instance)")
  ROW(1, 2, 4, "1 v_exp_f32 + 2 v_add_f32, 4 chains")
  ROW(2, 4, 2, "2 v_exp_f32 + 4 v_add_f32, 2 chains")
  // round 4: would ONE wave per SIMD (512 registers: 128 frames per wave, accumulators in AGPRs) keep the pipe fed?
  printf("one wave per SIMD (4-wave workgroups, one per CU):\n");
#define ROW1(E, A, CH, label)                                                          \
  {                                                                                    \
    const double t1 = run<E, A, CH>(1, cus, d_pat, d_out, target_ms);                  \
    printf("%-58s %7.0f TF   (%.2f of 2500)\n", label, t1, t1 / 2500);                 \
  }
  ROW1(0, 0, 4, "nothing, 4 chains")
  ROW1(0, 0, 2, "nothing, 2 chains")
  ROW1(1, 2, 4, "1 v_exp_f32 + 2 v_add_f32, 4 chains")
  ROW1(1, 3, 4, "1 v_exp_f32 + 3 v_add_f32, 4 chains (+ an accumulator read)")
  ROW1(1, 3, 2, "1 v_exp_f32 + 3 v_add_f32, 2 chains")
  return 0;
}
