// scatter_probe.hip -- what would the state sums of a WORK-SKIPPING clustered pass cost?  (round-5 review, item 4.)
// In a cluster-major exact pass a workgroup owns (cluster c, 64 of the frames that selected c) and holds, per frame, the
// partial sums of the ~30 states c's ~50 Gaussians belong to; they have to reach out[frame][state].  This probe issues
// exactly that traffic -- frames x 194 selected clusters x 30 states float atomic adds, a wave's 64 lanes on 64 different
// frame rows -- and nothing else (no matrix work, no gathers of the frames), for F = 200 000 frames; the figure scales
// linearly to 10^6.  Variant 2: plain stores of the same values into a compact [frames x 194 x 30] buffer (a segmented
// second pass would then read it back).      hipcc --offload-arch=gfx950 -O3 -o scatter_probe scatter_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void k_scatter(float *out, int64_t F, int S, const int *sel, int nsel, const int *cstate,
                                                 int per) {
  // block = (64-frame chunk x 4 waves: 4 of the frames' selected clusters per block step)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t f = (int64_t)blockIdx.x * 64 + lane;
  if (f >= F) return;
  for (int j = w; j < nsel; j += 4) {
    const int c = sel[(blockIdx.x * nsel + j)];   // (one list per chunk: frames of a chunk share their clusters here)
    const int *st = cstate + c * per;
    for (int k = 0; k < per; k++) atomicAdd(out + f * S + st[k], 1.0f + (float)k);
  }
}

__global__ __launch_bounds__(256) void k_compact(float *buf, int64_t F, int nsel, int per) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t f = (int64_t)blockIdx.x * 64 + lane;
  if (f >= F) return;
  for (int j = w; j < nsel; j += 4)
    for (int k = 0; k < per; k++) buf[((int64_t)(blockIdx.x * nsel + j) * per + k) * 64 + lane] = 1.0f + (float)k;
}

int main() {
  const int64_t F = 200000;
  const int S = 3136, C = 1000, nsel = 194, per = 30;
  float *out, *buf;
  int *sel, *cstate;
  hipMalloc(&out, F * S * 4);
  hipMalloc(&buf, (F / 64 + 1) * (int64_t)nsel * per * 64 * 4);
  hipMemset(out, 0, F * S * 4);
  std::vector<int> hsel((F / 64 + 1) * nsel), hcs(C * per);
  srand(7);
  for (auto &v : hsel) v = rand() % C;
  for (auto &v : hcs) v = rand() % 3125;
  hipMalloc(&sel, hsel.size() * 4);
  hipMalloc(&cstate, hcs.size() * 4);
  hipMemcpy(sel, hsel.data(), hsel.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(cstate, hcs.data(), hcs.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const unsigned blocks = (unsigned)((F + 63) / 64);
  float ms;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_scatter, dim3(blocks), dim3(256), 0, 0, out, F, S, sel, nsel, cstate, per);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("atomic scatter of %lld frames x %d clusters x %d states: %.2f ms -> %.1f ms per 10^6 frames (%.2e atomics)\n",
         (long long)F, nsel, per, ms, ms * 1e6 / F, (double)F * nsel * per);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_compact, dim3(blocks), dim3(256), 0, 0, buf, F, nsel, per);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("compact partial sums (coalesced stores, %.1f GB per 10^6 frames): %.2f ms -> %.1f ms per 10^6 frames, before a pass reads them back\n",
         1e6 * nsel * per * 4 / 1e9, ms, ms * 1e6 / F);
  return 0;
}
