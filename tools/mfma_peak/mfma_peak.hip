// Stand-alone matrix-pipe probe for gfx950: every wave issues v_mfma_f32_32x32x16_bf16 back to back on register
// operands (four independent accumulators, no memory traffic) for a fixed number of iterations; the host times
// the launch with HIP events and prints the executed bf16 TFLOP/s per operand pattern.  What it is for: the
// scoring kernel's ceiling under the board's power cap measured independently of that
// kernel -- and how much the operands' bit patterns matter (all-zero operands toggle nothing).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak [ms_per_launch]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                      \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// pattern 0: zeros; 1: random bf16 values in [1, 2) scaled by 2^-8 (random mantissas, one exponent);
// 2: random mantissas AND exponents over 2^-20 .. 2^4 (what a three-term split's low parts look like)
// experiment: does it matter whether consecutive instructions see operands of the same magnitude class?
// The registers a[k][.] / b[k][.] hold values around 2^0, 2^-8, 2^-16 (k = 0..2; random mantissas and signs) as the
// three terms of a split do, and both orders issue the six products of such a split, 48 instructions per pass.
template <bool GROUPED>
__global__ __launch_bounds__(256) void k_mfma_classes(const unsigned *__restrict__ pat, float *__restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  // [class][copy]: two different registers per magnitude class and side, so that consecutive instructions of the
  // grouped order still see different operands
  u32x4 a[3][2], b[3][2];
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int c = 0; c < 2; c++) {
      a[k][c] = ((const u32x4 *)pat)[((t * 16 + 2 * k + c) & 4095) * 4 + k];      // class k of the pattern buffer
      b[k][c] = ((const u32x4 *)pat)[((t * 16 + 8 + 2 * k + c) & 4095) * 4 + k];
    }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
#define MM(acc, ka, ca, kb, cb) \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ka][ca]), __builtin_bit_cast(bf16x8, b[kb][cb]), acc, 0, 0, 0)
  if (GROUPED) {
    // the six products of a three-term split, each for 8 instructions in a row (as a kernel that runs one product
    // over all K slabs and both row blocks before the next): 48 instructions per pass
    for (int i = 0; i < iters / 3; i++) {
#pragma unroll
      for (int q = 0; q < 2; q++) { MM(c0, 2, 0, 0, 0); MM(c1, 2, 1, 0, 1); MM(c2, 2, 0, 0, 1); MM(c3, 2, 1, 0, 0); }
#pragma unroll
      for (int q = 0; q < 2; q++) { MM(c0, 1, 0, 1, 0); MM(c1, 1, 1, 1, 1); MM(c2, 1, 0, 1, 1); MM(c3, 1, 1, 1, 0); }
#pragma unroll
      for (int q = 0; q < 2; q++) { MM(c0, 1, 0, 0, 0); MM(c1, 1, 1, 0, 1); MM(c2, 1, 0, 0, 1); MM(c3, 1, 1, 0, 0); }
#pragma unroll
      for (int q = 0; q < 2; q++) { MM(c0, 0, 0, 2, 0); MM(c1, 0, 1, 2, 1); MM(c2, 0, 0, 2, 1); MM(c3, 0, 1, 2, 0); }
#pragma unroll
      for (int q = 0; q < 2; q++) { MM(c0, 0, 0, 1, 0); MM(c1, 0, 1, 1, 1); MM(c2, 0, 0, 1, 1); MM(c3, 0, 1, 1, 0); }
#pragma unroll
      for (int q = 0; q < 2; q++) { MM(c0, 0, 0, 0, 0); MM(c1, 0, 1, 0, 1); MM(c2, 0, 0, 0, 1); MM(c3, 0, 1, 0, 0); }
    }
  } else {
    // the same 48 instructions, the product changing with every group of four (as a kernel that finishes a K slab
    // -- six products on four accumulators -- before the next)
    for (int i = 0; i < iters / 3; i++) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        MM(c0, 2, 0, 0, 0); MM(c1, 2, 1, 0, 1); MM(c2, 2, 0, 0, 1); MM(c3, 2, 1, 0, 0);
        MM(c0, 1, 0, 1, 0); MM(c1, 1, 1, 1, 1); MM(c2, 1, 0, 1, 1); MM(c3, 1, 1, 1, 0);
        MM(c0, 1, 0, 0, 0); MM(c1, 1, 1, 0, 1); MM(c2, 1, 0, 0, 1); MM(c3, 1, 1, 0, 0);
        MM(c0, 0, 0, 2, 0); MM(c1, 0, 1, 2, 1); MM(c2, 0, 0, 2, 1); MM(c3, 0, 1, 2, 0);
        MM(c0, 0, 0, 1, 0); MM(c1, 0, 1, 1, 1); MM(c2, 0, 0, 1, 1); MM(c3, 0, 1, 1, 0);
        MM(c0, 0, 0, 0, 0); MM(c1, 0, 1, 0, 1); MM(c2, 0, 0, 0, 1); MM(c3, 0, 1, 0, 0);
      }
    }
  }
#undef MM
  float s = 0;
#pragma unroll
  for (int e = 0; e < 16; e++) s += c0[e] + c1[e] + c2[e] + c3[e];
  out[t] = s;
}

__global__ __launch_bounds__(256) void k_mfma(const unsigned *__restrict__ pat, float *__restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  u32x4 a[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    a[k] = ((const u32x4 *)pat)[(t * 8 + k) & 16383];
    b[k] = ((const u32x4 *)pat)[(t * 8 + 4 + k) & 16383];
  }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const bf16x8 a0 = __builtin_bit_cast(bf16x8, a[r]), a1 = __builtin_bit_cast(bf16x8, a[(r + 1) & 3]);
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, b[r]), b1 = __builtin_bit_cast(bf16x8, b[(r + 2) & 3]);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c3, 0, 0, 0);
    }
  }
  float s = 0;
#pragma unroll
  for (int e = 0; e < 16; e++) s += c0[e] + c1[e] + c2[e] + c3[e];
  out[t] = s;
}

static void launch(int pattern, int blocks, const unsigned *d_pat, float *d_out, int iters) {
  // every kernel issues 16 instructions per wave and iteration of `iters`
  if (pattern == 3) hipLaunchKernelGGL(k_mfma_classes<false>, dim3(blocks), dim3(256), 0, 0, d_pat, d_out, iters);
  else if (pattern == 4) hipLaunchKernelGGL(k_mfma_classes<true>, dim3(blocks), dim3(256), 0, 0, d_pat, d_out, iters);
  else hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d_pat, d_out, iters);
}

static unsigned short bf16_of(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main(int argc, char **argv) {
  const double target_ms = argc > 1 ? atof(argv[1]) : 40.0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 2;  // 8 waves per CU, two per SIMD
  unsigned *d_pat;
  float *d_out;
  CHECK(hipMalloc(&d_pat, 16384 * 16));
  CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%s, %d CUs, %d waves per CU, v_mfma_f32_32x32x16_bf16 (32768 flop per wave and instruction)\n", prop.name, cus, 8);
  const char *names[5] = {"all-zero operands", "random mantissas, one exponent", "random mantissas and exponents",
                          "split-term classes, mixed order", "split-term classes, grouped order"};
  const int n_patterns = getenv("MFMA_PEAK_CLASSES") ? 5 : 3;
  for (int pattern = 0; pattern < n_patterns; pattern++) {
    std::vector<unsigned short> h(16384 * 8);
    srand(1234);
    for (size_t i = 0; i < h.size(); i++) {
      if (pattern == 0) h[i] = 0;
      else if (pattern >= 3) {  // u32x4 element e of the buffer belongs to class e % 4: 2^0, 2^-8, 2^-16, 2^0
        const int cls = (int)((i / 8) % 4);
        const float m = 1.0f + (float)(rand() & 127) / 128.0f;
        const int ex = (cls == 1 ? -8 : cls == 2 ? -16 : 0) + (rand() % 5) - 2;
        h[i] = bf16_of((rand() & 1 ? -m : m) * (float)ldexp(1.0, ex));
      } else {
        const float m = 1.0f + (float)(rand() & 127) / 128.0f;
        const int ex = pattern == 1 ? -8 : (rand() % 25) - 20;
        h[i] = bf16_of((rand() & 1 ? -m : m) * (float)ldexp(1.0, ex));
      }
    }
    CHECK(hipMemcpy(d_pat, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    int iters = 2000;
    for (int rep = 0; rep < 2; rep++) {  // calibrate the iteration count to the target duration
      CHECK(hipEventRecord(e0));
      launch(pattern, blocks, d_pat, d_out, iters);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      iters = ((int)(iters * target_ms / ms) + 48) / 48 * 48;
    }
    double best = 0, sum = 0;
    const int reps = 12;
    for (int rep = 0; rep < reps; rep++) {
      CHECK(hipEventRecord(e0));
      launch(pattern, blocks, d_pat, d_out, iters);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double tf = (double)blocks * 4 * iters * 16.0 * 32768.0 / (ms * 1e-3) / 1e12;
      if (rep >= 2) {  // the first launches run before the clock has settled under load
        sum += tf;
        if (tf > best) best = tf;
      }
    }
    printf("%-36s %7.0f TFLOP/s mean of %d launches of ~%.0f ms (best %.0f) = %.2f of 2500\n", names[pattern], sum / (reps - 2),
           reps - 2, target_ms, best, sum / (reps - 2) / 2500.0);
  }
  return 0;
}
